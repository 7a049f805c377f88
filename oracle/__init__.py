"""CPU checkers for the batched-MCTS hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under this directory is part of the product: only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  PARITY UNPINNED (no reference tests / golden vectors exist and
the reference cannot be run in the build environment); see mcts_oracle.h.
"""
