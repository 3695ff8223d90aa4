"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatements (torch-CPU fp32 + a plain-C index-stream generator) of the reference's
algorithm for the hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything from here; the shipped product path
(video-to-action-release_amd/) never does and fails loudly if libv2a_hip.so is missing.

Parity pinning (DESIGN.md "Oracle"):
  * in-tree arithmetic (UNetModel, Conv3d, attention, Perceiver, GoalGaussianDiffusion,
    ConditionalUnet1D, SpatialSoftmax, normaliser, replay buffer index stream) is PINNED against
    outputs of the reference itself, imported in the build container by tools/make_golden.py
    and committed as tests/golden/*.npz.
  * arithmetic living in third-party packages that are absent from /root/reference
    (diffusers DDPM/DDIM schedulers [unpinned version], torchvision 0.15.1 resnet18 topology,
    ema_pytorch 0.2.3 EMA, torch AdamW / clip_grad_norm_) is restated from the published
    algorithms: PARITY UNPINNED by any reference test (the reference has none); anchored on the
    reference's call sites only.  numpy's legacy RandomState.randint/uniform and CPython's
    random.randint ARE pinned: replay_ref.c is checked bit-for-bit against numpy/CPython.
"""
