#!/bin/bash
# Run on the GPU box (through gpurun): every measurement the round's profiles/ directory holds.
#   tools/profile_round.sh r02
# rocprofv3 passes follow MI355X_MICROARCH.md: counters in their own runs (one group per pass), never combined with
# --sys-trace / runtime traces; the program itself after `--` (python3, no env/bash hop).
set -e -o pipefail
TAG=${1:-rXX}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" profiles
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_trace" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --batched-probe 0 > "$OUT/bench_trace.log" 2>&1
echo "bench trace done"
export SFMX_PROF_META="$PWD/profiles/${TAG}_pmc_meta.json"
export SFMX_PROF_C4=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pmc/trace" -- python3 tools/prof_kernels.py > "$OUT/pmc_trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/fetch" -- python3 tools/prof_kernels.py > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/write" -- python3 tools/prof_kernels.py > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc/sq" -- python3 tools/prof_kernels.py > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d "$OUT/pmc/mfma" -- python3 tools/prof_kernels.py > "$OUT/pmc_mfma.log" 2>&1 || echo "mfma counter pass not available"
echo "pmc passes done"
python tools/summarize_profiles.py "$TAG" "$OUT/bench_trace" "$OUT/pmc"
# the bench line last: its roofline object quotes the PMC summary written just above (same code, same box)
python bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.err"
cp "$OUT/bench_line.json" "profiles/${TAG}_bench_line.json"
echo "bench line done"
python tools/microbench.py > "profiles/${TAG}_microbench.txt" 2>&1
echo "microbench done"
python tools/bench_c3.py --frames ${C3_FRAMES:-1000} --passes ${C3_PASSES:-2} > "profiles/${TAG}_c3_line.json" 2> "$OUT/c3.err"
echo "c3 done"
python tools/bench_c5.py --frames ${C5_FRAMES:-60} > "profiles/${TAG}_c5_prefix_line.json" 2> "$OUT/c5.err"
echo "c5 done"
python bench.py --mode ba-sharded --steps 10 --warmup 2 > "profiles/${TAG}_ba_sharded_1gpu_line.json" 2> "$OUT/ba_sharded.err"
python tools/posegraph_c5.py > "profiles/${TAG}_posegraph_c5.txt" 2> "$OUT/pg.err"
python tools/ransac_cond_probe.py > "profiles/${TAG}_ransac_cond_probe.txt" 2> "$OUT/cond.err"
PROBE_FRAMES=47 python tools/ransac_cond_probe.py 2> "$OUT/cond_all.err" | grep -v "^n=\|^cond\|^  cond\|^rel\|^uncertain\|^device rows" > "profiles/${TAG}_ransac_cond_all_pairs.txt"
python tools/virtual_world_probe.py > "profiles/${TAG}_virtual_world.jsonl" 2> "$OUT/vw.err"
SFMX_KLT_STAMPS=1 python tools/klt_stamps.py 2>&1 | grep -v amdgpu.ids > "profiles/${TAG}_klt_stamps.txt"
python tools/klt_variant_probe.py 2>&1 | grep -v amdgpu.ids > "profiles/${TAG}_klt_variants_probe.txt"
echo "extras done"
mkdir -p "gpurun_out/profiles_$TAG" && cp profiles/${TAG}_* "gpurun_out/profiles_$TAG/"
echo "profiles written: $(ls profiles | grep "^$TAG" | tr '\n' ' ')"
