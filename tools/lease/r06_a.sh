#!/bin/bash
# round 6, lease A: (1) real-RCCL self-exchange floor at world = 1, (2) interleaved A/B of the wave-local sweeps (ECFFT_WAVE_LOCAL 0 / 1 / 2)
# and of the batch split (ECFFT_BATCH_SPLIT 0 / 1), (3) the whole GPU suite on the new default build, (4) the driver's SCALE command
# with N = 8 ranks sharing the one GPU over the RCCL stand-in (functional only)
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
(time timeout 300 python tools/rccl_self_exchange.py $O/rccl_self_exchange.txt) > $O/rccl.log 2>&1; tail -3 $O/rccl.log
{
for ln in 20 18; do echo "== secp256k1 2^$ln  (wl0 = workgroup barriers everywhere, libecfft_hip = wavefront fence, wl2 = s_waitcnt only)"; python tools/ab_many.py secp256k1 $ln $V/wl0.so ecfft_amd/libecfft_hip.so $V/wl2.so 2>&1 | tail -3; done
echo "== secp256k1 2^20 x 8 (batched)"; python tools/ab_many.py secp256k1 20 --count 8 $V/wl0.so ecfft_amd/libecfft_hip.so $V/wl2.so 2>&1 | tail -3
} > $O/wave_local_ab.txt 2>&1
cat $O/wave_local_ab.txt
{
echo "== batch split: bs0 = one stream (rounds 1-5), libecfft_hip = two half-batches on two streams; ms per polynomial pair"
for cfg in "20 8" "20 2" "19 4" "18 8" "20 4"; do set -- $cfg; echo "-- secp256k1 2^$1 x $2"; python tools/ab_many.py secp256k1 $1 --count $2 $V/bs0.so ecfft_amd/libecfft_hip.so 2>&1 | tail -2; done
echo "-- m31 2^24 x 2"; python tools/ab_many.py m31 24 --count 2 $V/bs0.so ecfft_amd/libecfft_hip.so 2>&1 | tail -2
} > $O/batch_split_ab.txt 2>&1
cat $O/batch_split_ab.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/suite.log 2>&1; tail -8 $O/suite.log
# the driver's exact SCALE command, N = 8
N=8
(time ECFFT_BENCH_BACKEND=gloo ECFFT_BENCH_TRANSPORT=rccl ECFFT_BENCH_RCCL_LIB=$PWD/tests/stub_rccl/librccl_stub.so MASTER_ADDR=127.0.0.1 timeout 1200 \
  python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29508 bench.py --gpus $N --steps 3 --warmup 1 2> $O/err_$N.log | grep "^{" > $O/scale_$N.json) 2> $O/time_$N.log
tail -3 $O/time_$N.log
python - <<PY
import json
d = json.load(open("$O/scale_$N.json"))
sp = d["split"]
print("N=$N headline", d.get("headline"), "scaling", d["scaling"], "ms_per_step", round(d["ms_per_step"], 3), "split status", sp["status"], "exchanges/step", sp["enter_exit"]["phases"]["exchanges_per_step"],
      "round trip", sp["enter_exit"]["round_trip_ok"], "ranks seen", sp["enter_exit"]["ranks_seen_by_transport"], "extend ok", sp["extend"]["round_trip_ok"], "extend exchanges/step", sp["extend"]["phases"]["exchanges_per_step"],
      "replicas ms", round(d["replicas"]["ms_per_step"], 3))
PY
tail -3 $O/err_$N.log
