"""End-to-end training steps on the GPU: self-play search -> targets -> replay -> SGD."""
import math

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy,use_gae,atoms,per_alpha", [("gumbel", False, 1, 0.0),
                                                           ("sampled", True, 4, 0.6),
                                                           ("muzero", False, 1, -2.0)])
def test_training_steps_run_and_learn_signal(policy, use_gae, atoms, per_alpha):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(num_envs=64, unroll_length=6, num_simulations=16, policy=policy, use_gae=use_gae,
                 num_atoms=atoms, per_alpha=per_alpha, max_replay_size=2048,
                 replay_buffer_batch_size=32, num_minibatches=4, value_hidden=(64, 64),
                 policy_hidden=(32, 32), num_samples=8)
    tr = Trainer(cfg, device="cuda")
    before = [p.clone() for p in tr.optimizer.params]
    m = tr.train(3)
    assert all(math.isfinite(v) for v in m.values()), m
    assert m["buffer_current_size"] == 3 * 6 * 64 and m["training/sps"] > 0
    assert tr.optimizer.count == 3 * cfg.num_minibatches
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.optimizer.params))
    assert tr.stats.count == 3 * 6 * 64
    # the self-play model sees the updated parameters and statistics
    lg, _ = tr.model.forward(tr.env_state.obs)
    from alphazero_amd import learner as L
    ref, _ = L.forward_networks(tr.policy_layers, tr.value_layers, tr.stats.mean, tr.stats.std,
                                tr.env_state.obs)
    torch.testing.assert_close(lg, ref, rtol=1e-4, atol=1e-5)
    ev = tr.evaluate(4)
    assert ev["mean_reward"] > 0
