#!/bin/bash
# round 5: dry run of the driver's SCALE command at its REAL sizes with 2 and 4 ranks sharing the one GPU: gloo process group, the
# RCCL transport bound to the stand-in library (tests/stub_rccl) — functional only, the times mean nothing
O=gpurun_out/r05scale; rm -rf $O; mkdir -p $O
for N in 2 4; do
  ECFFT_BENCH_BACKEND=gloo ECFFT_BENCH_TRANSPORT=rccl ECFFT_BENCH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so MASTER_ADDR=127.0.0.1 timeout 900 \
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 3 --warmup 1 2> $O/err_$N.log | grep "^{" > $O/scale_$N.json
  python - <<PY
import json
d = json.load(open("$O/scale_$N.json"))
print("N=$N headline", d.get("headline"), "scaling", d["scaling"], "ms_per_step", round(d["ms_per_step"], 3), "split status", d["split"]["status"], "exchanges/step", d["split"]["enter_exit"]["phases"]["exchanges_per_step"],
      "round trip", d["split"]["enter_exit"]["round_trip_ok"], "extend ok", d["split"]["extend"]["round_trip_ok"], "replicas ms", round(d["replicas"]["ms_per_step"], 3))
PY
done
tail -3 $O/err_4.log
