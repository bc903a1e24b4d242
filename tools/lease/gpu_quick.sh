#!/bin/bash
# quick GPU check: parity tests (fast subset unless FULL=1) + the two bench lines
mkdir -p gpurun_out/q
if [ "${FULL:-0}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/q/pytest.log 2>&1
else
  timeout 900 python -m pytest tests -m gpu -x -q -k "not 2e18 and not hip_ops and not 2e22" > gpurun_out/q/pytest.log 2>&1
fi
echo "pytest rc=$?" >> gpurun_out/q/pytest.log
tail -15 gpurun_out/q/pytest.log
for f in secp256k1:20 m31:24; do
  fld=${f%%:*}; ln=${f##*:}
  timeout 600 python bench.py --field $fld --log-n $ln --cpu-log-n 0 > gpurun_out/q/bench_$fld.json 2> gpurun_out/q/bench_$fld.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/q/bench_$fld.json").read().strip().splitlines()[-1])
    print("$fld", "ms_per_step", round(d["ms_per_step"],3), "enter", round(d.get("enter_ms",0),3), "exit", round(d.get("exit_ms",0),3), "batched", d["batched"] and round(d["batched"]["ms_per_transform_pair"],3))
    for k in d["roofline"]["per_class"]: print("   ", k["name"], k["launches_per_step"], round(k["event_ms_per_step"],3), "hbm_frac", k.get("hbm_frac"))
except Exception as e:
    print("$fld bench failed", e); print(open("gpurun_out/q/bench_$fld.err").read()[-2000:])
PY
done
