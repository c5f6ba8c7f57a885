#!/bin/sh
# round 6, call 5: gather_mfma2_kernel, second cut (three stage buffers, features from memory, contiguous tap stores): bits, time, probes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout 300 python tools/gather_dump.py /tmp/v2.pt 2>&1 | tail -1
PIPS_LIB_PATH=$R/build/libpips_gmv1.so timeout 300 python tools/gather_dump.py /tmp/v1.pt 2>&1 | tail -1
python tools/gather_dump.py --compare /tmp/v1.pt /tmp/v2.pt 2>&1 | head -8
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "mfma or bf16_maps or tiled" 2>&1 | tail -3
timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16\|config-3"
for a in 8 10; do
  sh tools/build_gather_variant.sh g2a$a -DG2_ABLATE=$a > /dev/null 2>&1
  echo "== G2_ABLATE=$a"
  PIPS_LIB_PATH=$R/build/libpips_g2a$a.so timeout 300 python tools/gather_c4.py 2>&1 | grep "bf16" | sed -e 's/.*gather_mfma_kernel)://'
done
sh tools/build_gather_variant.sh g2tr -DG2_TRACE > /dev/null 2>&1
PIPS_LIB_PATH=$R/build/libpips_g2tr.so timeout 300 python tools/g2_trace.py 2>&1 | grep -v amdgpu.ids
