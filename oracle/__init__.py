"""CPU oracle for the RLControl hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (rlcontrol_amd/) never does; it fails loudly when its HIP library is missing.

Contents
  ddpg_oracle.c / liboracle.so   fp32 C restatement of DDPG update_network (agents/DDPG.py:74-95)
  ddpg.py                        ctypes wrapper + parameter-blob helpers + TF-style initialisers
  cpu_baseline.py                reference-structured CPU loop (list-of-records replay, sample_n_k,
                                 float64 TD glue) used for bench.py's "cpu_baseline" number
Parity status: numpy-side behaviour is pinned by tests/golden (generated from the reference);
the TensorFlow-1.15 arithmetic is "parity unpinned" (third-party, absent) and is cross-checked by an
independent float64 autograd restatement in tests/torch_ref.py.
"""
