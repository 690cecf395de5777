"""CPU oracle for the FastSpeech2 training hot path  --  TEST INFRASTRUCTURE ONLY.

This package restates, in plain eager PyTorch on the CPU (fp32, or fp64 on request), the
algorithm of the reference path ``train_fastspeech2.py`` -> ``Models.fastspeech2.FastSpeech2``
(syoamakase/Transformer_TTS).  Every function cites the reference file:line it follows.

* Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
  it, and only as the checker / the timed CPU baseline.  The product (``transformer_tts_amd``)
  never imports it and has no CPU fallback: it raises when ``libfs2_hip.so`` is missing.
* Pinning: the restatement is checked against golden vectors produced by importing the real
  reference in the build container (``tests/golden/make_golden.py`` + ``tests/golden/*.npz``):
  all 9 non-None forward outputs, the 5 losses, every parameter gradient and the parameters /
  BatchNorm buffers after 1 and 3 optimizer steps of the reference's own ``train_loop``
  (``tests/test_oracle_golden.py``).  The reference ships no tests or known-answer vectors of its
  own, so these generated vectors are the pin.
* Arithmetic library: the reference's numerics are PyTorch's (requirements.txt pins the stale
  torch==1.5.0; the code needs >= 1.6); the oracle runs on torch 2.10.0+rocm7.0 CPU as installed.
"""
