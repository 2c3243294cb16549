#!/bin/bash
# (branch next/hip-fixtures) the reference-code fixtures of round 3 on the real library: the sampler loops (SD1.5, SDXL,
# attribute concentration), the mask-loss assembly, the third-party VAE decoder and the whole step of the reference's loop
# body - stand-in networks as torch ops on the GPU; fused CFG + DDPM step, attention-map gather, discriminator head, clip +
# AdamW kernels through the C ABI.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_models.py tests/test_losses.py tests/test_step.py -m gpu -q -p no:cacheprovider \
  -k "reference or third_party" > gpurun_out/r4h_fixtures.log 2>&1; tail -15 gpurun_out/r4h_fixtures.log
