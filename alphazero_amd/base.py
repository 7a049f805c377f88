"""Core types of the policy API (mirrors mctx_dist/_src/base.py).

Arrays are torch tensors on the GPU.  `value` may be `[B]` (the scalar layout
of mctx/_src/base.py) or `[B, num_atoms]` (mctx_dist/_src/base.py); the layout
of the returned Tree follows the root's.
"""
import dataclasses
from typing import Any, Callable, Generic, Tuple, TypeVar

T = TypeVar("T")


class _Replaceable:
    def replace(self, **changes):
        return dataclasses.replace(self, **changes)


@dataclasses.dataclass(frozen=True)
class RecurrentFnOutput(_Replaceable):
    """mctx_dist/_src/base.py:34-46: reward [B], discount [B], prior_logits [B,A], value [B(,K)]."""
    reward: Any
    discount: Any
    prior_logits: Any
    value: Any


@dataclasses.dataclass(frozen=True)
class RootFnOutput(_Replaceable):
    """mctx_dist/_src/base.py:57-66: prior_logits [B,A], value [B(,K)], embedding pytree [B,...]."""
    prior_logits: Any
    value: Any
    embedding: Any


@dataclasses.dataclass(frozen=True)
class PolicyOutput(_Replaceable, Generic[T]):
    """mctx_dist/_src/base.py:87-99: action [B] int32, action_weights [B,A], search_tree."""
    action: Any
    action_weights: Any
    search_tree: Any


@dataclasses.dataclass(frozen=True)
class DecisionRecurrentFnOutput(_Replaceable):
    """mctx_dist/_src/base.py:103-118: chance_logits [B,C], afterstate_value [B]."""
    chance_logits: Any
    afterstate_value: Any


@dataclasses.dataclass(frozen=True)
class ChanceRecurrentFnOutput(_Replaceable):
    """mctx_dist/_src/base.py:121-139: action_logits [B,A], value [B], reward [B], discount [B]."""
    action_logits: Any
    value: Any
    reward: Any
    discount: Any


@dataclasses.dataclass(frozen=True)
class StochasticRecurrentState(_Replaceable):
    """mctx_dist/_src/base.py:142-161: the embedding of a Stochastic MuZero node; a decision
    node's afterstate_embedding and a chance node's state_embedding are dummies."""
    state_embedding: Any
    afterstate_embedding: Any
    is_decision_node: Any  # [B] bool


Params = Any
RecurrentState = Any
RecurrentFn = Callable[[Params, Any, Any, RecurrentState], Tuple[RecurrentFnOutput, RecurrentState]]
