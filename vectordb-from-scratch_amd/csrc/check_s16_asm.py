#!/usr/bin/env python3
"""Build-time check of kernels_fused_s16.hip's generated code (called by the Makefile).

The kernel keeps its 256 accumulators in a[0:255] BY HAND (inline asm on fixed AccVGPRs); hipcc does not know that and would
use "free" AccVGPRs as spill space the moment register pressure rises -- silently overwriting accumulators.  So every
instance must come out of the compiler with
  * no scratch (a scratch reload is also a vector-memory operation that drains the DMA pipeline),
  * no AccVGPR access the compiler made up: every v_accvgpr_read/_write must be one of ours (the a[N] spelling of the asm
    strings; compiler-generated ones print as aN), and no v_accvgpr_mov.
Usage: check_s16_asm.py kernels_fused_s16.s"""
import re
import sys

txt = open(sys.argv[1]).read()
bad = []
kernels = re.findall(r"^(_ZN3vdb16fused_s16_kernel\w+):", txt, flags=re.M)
if len(kernels) < 3:
    bad.append(f"expected 3 kernel instances, found {len(kernels)}")
for name in kernels:
    body = txt[txt.index(name + ":"):]
    body = body[:body.index("s_endpgm")]
    n_scratch = len(re.findall(r"^\s+scratch_", body, flags=re.M))
    made_up = re.findall(r"^\s+v_accvgpr_(?:write_b32 a\d|read_b32 v\d+, a\d|mov)", body, flags=re.M)
    ours_r = len(re.findall(r"^\s+v_accvgpr_read_b32 v\d+, a\[", body, flags=re.M))
    n_mfma = len(re.findall(r"^\s+v_mfma_f32_32x32x16_bf16 a\[", body, flags=re.M))
    if n_scratch:
        bad.append(f"{name}: {n_scratch} scratch instructions")
    if made_up:
        bad.append(f"{name}: {len(made_up)} compiler-generated AccVGPR accesses")
    if ours_r not in (256, 512) or n_mfma not in (64, 96):           # 512: the filter epilogue exists twice (with / without the NaN test); 96: the diagnostics build's compute-only k-step 1
        bad.append(f"{name}: {ours_r} accumulator reads (256 or 512 expected), {n_mfma} MFMAs (64 expected)")
if bad:
    print("kernels_fused_s16: generated code violates the hand-allocated AccVGPR contract:\n  " + "\n  ".join(bad), file=sys.stderr)
    sys.exit(1)
print(f"kernels_fused_s16: {len(kernels)} instances, no scratch, no compiler-made AccVGPR access")
