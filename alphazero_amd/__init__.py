"""MI355X-native batched MCTS behind the mctx policy API.

``import alphazero_amd as mctx`` gives the names the reference's driver uses from
``mctx_dist`` (mctx_dist/__init__.py): the three policies, the API types, the
qtransforms and the selection functions.  All compute runs in the HIP kernels of
``libaz_mcts.so``; importing the package does not load the library, the first
call does, and it raises if the library is missing (there is no fallback path).
"""
from .action_selection import (GumbelMuZeroExtraData, gumbel_muzero_interior_action_selection,
                               gumbel_muzero_root_action_selection, muzero_action_selection,
                               sampled_muzero_action_selection)
from .base import (ChanceRecurrentFnOutput, DecisionRecurrentFnOutput, PolicyOutput, RecurrentFnOutput,
                   RootFnOutput, StochasticRecurrentState)
from .networks import FusedModel, FusedRecurrentFn, init_mlp
from .policies import (actor_post, gumbel_muzero_policy, muzero_policy, sampled_muzero_policy,
                       stochastic_muzero_policy)
from .qtransforms import (qtransform_by_min_max, qtransform_by_parent_and_siblings,
                          qtransform_completed_by_mix_value)
from .search import InjectedRecurrentFn, search
from .seq_halving import get_sequence_of_considered_visits, get_table_of_considered_visits
from .tree import SearchSummary, Tree, infer_batch_size
from . import envs
from . import minatar
from . import replay_buffers

__version__ = "0.1.0"

__all__ = [
    "GumbelMuZeroExtraData", "PolicyOutput", "RecurrentFnOutput", "RootFnOutput", "SearchSummary",
    "Tree", "FusedModel", "FusedRecurrentFn", "InjectedRecurrentFn", "actor_post", "envs",
    "get_sequence_of_considered_visits", "get_table_of_considered_visits",
    "gumbel_muzero_interior_action_selection", "gumbel_muzero_policy",
    "gumbel_muzero_root_action_selection", "infer_batch_size", "init_mlp",
    "muzero_action_selection", "muzero_policy", "qtransform_by_min_max",
    "qtransform_by_parent_and_siblings", "qtransform_completed_by_mix_value",
    "sampled_muzero_action_selection", "sampled_muzero_policy", "search",
    "ChanceRecurrentFnOutput", "DecisionRecurrentFnOutput", "StochasticRecurrentState",
    "stochastic_muzero_policy", "minatar",
]
