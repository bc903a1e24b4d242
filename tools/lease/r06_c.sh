#!/bin/bash
# round 6, lease C: (1) the split projection re-run with the MEASURED exchange floors (15 us pairwise, 36 us for a 7-peer group) beside
# the assumed 25 us, at 2^20 and 2^22 (plain and striped); (2) the whole GPU suite (new: parser mutation loop, N = 8 dry run, oracle
# holes); (3) the driver's command with N = 8, n = 2^22 and the link striping switched on (functional, over the stand-in)
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
for d in 15 25 36; do
  timeout 900 python tools/split_project.py 20 $d 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e20_d$d.txt; tail -5 $O/split_projection_2e20_d$d.txt | cut -c1-220
  WORLDS=8 timeout 900 python tools/split_project.py 22 $d 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_d${d}_plain.txt; tail -9 $O/split_projection_2e22_d${d}_plain.txt | cut -c1-220
  WORLDS=8 ECFFT_STRIPE_MIN_GAIN=4194304 timeout 900 python tools/split_project.py 22 $d 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_d${d}_striped.txt; tail -9 $O/split_projection_2e22_d${d}_striped.txt | cut -c1-220
done
WORLDS=8 ECFFT_SPLIT_GATHER_MAX_LOG=22 timeout 900 python tools/split_project.py 22 15 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_d15_gather.txt
(time timeout 1800 python -m pytest tests -m gpu -x -q --durations=15) > $O/suite.log 2>&1; tail -30 $O/suite.log
N=8
(time ECFFT_BENCH_BACKEND=gloo ECFFT_BENCH_TRANSPORT=rccl ECFFT_BENCH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so MASTER_ADDR=127.0.0.1 timeout 1200 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus $N --steps 2 --warmup 1 --log-n 20 --split-log-n 22 --split-log-e 0 --stripe-min-gain 4194304 --batch 0 2> $O/err_stripe.log | grep "^{" > $O/scale_8_2e22_striped.json) 2> $O/time_stripe.log
(ECFFT_BENCH_BACKEND=gloo ECFFT_BENCH_TRANSPORT=rccl ECFFT_BENCH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so MASTER_ADDR=127.0.0.1 timeout 1200 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus $N --steps 2 --warmup 1 --log-n 20 --split-log-n 22 --split-log-e 0 --batch 0 2> $O/err_plain.log | grep "^{" > $O/scale_8_2e22_plain.json)
python - <<PY
import json
for tag in ("striped", "plain"):
    d = json.load(open("$O/scale_8_2e22_%s.json" % tag))
    sp = d["split"]; ee = sp["enter_exit"]
    print(tag, "| status", sp["status"], "| round trip", ee["round_trip_ok"], "| ranks seen", ee["ranks_seen_by_transport"], "| exchanges/step", ee["phases"]["exchanges_per_step"], "| MB sent/step/rank", round(ee["phases"]["bytes_sent_per_step_per_rank"] / 1e6, 1), "| split_exit", ee["config"].get("split_exit"))
PY
tail -3 $O/time_stripe.log; tail -2 $O/err_stripe.log | cut -c1-300
