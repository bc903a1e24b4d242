#!/bin/bash
# round 3: per-class counters of the two bench workloads -> gpurun_out/ctr_r03_*/counters.json
bash tools/prof_counters.sh r03_secp secp256k1 20 3 > /dev/null 2>&1
bash tools/prof_counters.sh r03_m31 m31 24 3 > /dev/null 2>&1
ls gpurun_out/ctr_r03_secp gpurun_out/ctr_r03_m31
python - <<'PY'
import json
for t in ("secp", "m31"):
    d = json.load(open(f"gpurun_out/ctr_r03_{t}/counters.json"))
    for k, v in d["classes"].items():
        print(t, k, {a: round(b, 1) for a, b in v.items() if a in ("launches_per_step", "avg_us", "pmc_avg_us", "SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_I8", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES", "hbm_bytes_per_launch")})
PY
