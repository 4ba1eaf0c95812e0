# A/B of the RLFM lane walk on config 4b (bench.py --workload rep-rlfm, measurement build): the rounds kernel (round 6)
# against round 5's fmx_locate_rl_lane_kernel, and the grid cap of the former
mkdir -p gpurun_out/r06
M=$PWD/fm_index_amd/libfmx_measure.so
run() {
  echo "$*"
  env FMX_LIB=$M "$@" timeout 600 python3 bench.py --workload rep-rlfm --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-census --no-d2h --no-accel --no-rccl-check --no-wide --no-ic-ab --no-config5 --no-early-exit --detail-out gpurun_out/r06/rlr_tmp.json 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('   ', {k: d.get(k) for k in ('locate_hits_per_s','locate_ms_per_batch')})"
}
for i in 1 2; do run FMX_RL_ROUNDS=1; run FMX_RL_ROUNDS=0; done
for B in 2048 8192 32768 131072; do run FMX_RL_ROUNDS=1 FMX_RL_ROUNDS_BLOCKS=$B; done
