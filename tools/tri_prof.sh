# Phase profile of the block-tridiagonal register kernel (s_memtime stamps accumulated per wave, printed by the kernel).
# Run on the GPU box (the snapshot there is scratch):  gpurun -- bash tools/tri_prof.sh
set -e
touch jrl-qp_amd/csrc/structured.hip
make -C jrl-qp_amd/csrc EXTRA=-DTRI_PROF > gpurun_out/tri_prof_build.log 2>&1
python tools/time_structured.py 2>&1 | grep -E "tri wave|tridiag" | tail -12
