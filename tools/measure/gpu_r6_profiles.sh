#!/bin/bash
# Round-6 profile set: bench lines (the default -- which carries the measured roofs and BASELINE config 5 itself --, the cluttered and the noisy
# scene, RCCL world of one), rocprofv3 kernel stats of the default and of the config-5 command, the rasteriser's traffic / L2 / SQ counters at the
# benchmark's launch shape and at config 5's (8192 renders per launch now).  tools/measure/refresh_profiles_r6.py turns the output into profiles/r06_*.
#   gpurun -- 'bash tools/measure/gpu_r6_profiles.sh' ; gpurun -- 'bash tools/measure/gpu_verifier_traffic.sh 4096 50 && bash tools/measure/gpu_verifier_traffic.sh 4096 152'
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6s
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$GRAFT_REPO_ROOT"
step() { local secs=$1 log=$2; shift 2; echo "== $*" | tee -a "$OUT/steps.log"; timeout -k 10 "$secs" "$@" > "$OUT/$log" 2>&1; local rc=$?; echo "   rc=$rc" | tee -a "$OUT/steps.log"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo TIMEOUT | tee -a "$OUT/steps.log"; exit 1; fi; return 0; }
C5="--pano-hw 1024x2048 --surfaces floor,ceiling --layers 152 --panos 16"
LEAN="--no-cpu-baseline --no-calibration --no-config5"
step 500 bench.log python bench.py --steps 20 --warmup 5
step 300 bench_cluttered.log python bench.py --steps 10 --warmup 3 --scene cluttered $LEAN
step 300 bench_noisy.log python bench.py --steps 10 --warmup 3 --scene noisy $LEAN
step 400 bench_c5.log python bench.py $C5 --steps 5 --warmup 2 $LEAN
step 300 bench_rccl.log python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --force-dist $LEAN
cd /tmp
step 300 prof1.log rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof1" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-power-probe $LEAN
step 400 prof5.log rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof5" -- python3 "$GRAFT_REPO_ROOT/bench.py" $C5 --steps 2 --warmup 1 --no-power-probe $LEAN
R="$GRAFT_REPO_ROOT/tools/measure/pmc_render.py"
step 300 pmc_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $R 4096 64
step 300 pmc_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $R 4096 64
step 300 pmc_h.log rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_hit" -- python3 $R 4096 64
step 300 dsq.log rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/dsq" -- python3 $R 4096 64
step 400 pmc5_f.log rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc5_fetch" -- python3 $R 8192 16 1024x2048 floor,ceiling
step 400 pmc5_w.log rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc5_write" -- python3 $R 8192 16 1024x2048 floor,ceiling
find "$OUT" -name "*.db" -delete
P="$GRAFT_REPO_ROOT/tools/measure/pmc_report.py"
for p in pmc_fetch pmc_write pmc_hit; do python3 $P "$OUT/$p" bev_; done > "$OUT/ras_traffic.txt"
for p in pmc5_fetch pmc5_write; do python3 $P "$OUT/$p" bev_; done > "$OUT/ras_traffic_config5.txt"
python3 $P "$OUT/dsq" bev_ > "$OUT/ras_sq.txt"
for f in bench bench_cluttered bench_noisy bench_c5 bench_rccl; do echo $f; grep '^{' "$OUT/$f.log" | tail -1 | cut -c1-160; done
cat "$OUT/ras_traffic.txt" "$OUT/ras_traffic_config5.txt"
