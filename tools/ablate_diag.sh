# diagnostic build (-DVDB_DIAG) with staging stores / loads switchable at run time; restores the normal build after
set -e
cd vectordb-from-scratch_amd/csrc && make clean >/dev/null && make -j4 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DVDB_DIAG" >/dev/null 2>&1 && cd ../..
for a in 0 8 40 72 104; do echo -n "ablate=$a kernel_ms,step_ms: "; VDB_FUSED_ABLATE=$a timeout -k 10 120 python bench.py --steps 5 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['roofline']['kernel_ms'], d['ms_per_step'])"; done
cd vectordb-from-scratch_amd/csrc && make clean >/dev/null && make -j4 >/dev/null 2>&1
