for v in 4 6008 6016 6024; do SQ_BWD="--bwd $v" bash scripts/gpu_sq_counters.sh > gpurun_out/r06_f_sq_$v.log 2>&1; cp gpurun_out/sq_counters.json gpurun_out/r06_f_sq_$v.json; python - <<PY
import json
d=json.load(open("gpurun_out/r06_f_sq_$v.json")).get("corr_bwd_f16x2",{})
print("variant $v:", {k: round(d.get(k,0)) for k in ("SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_INSTS_LDS","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_LDS","SQ_BUSY_CYCLES","SQ_INSTS_VALU","SQ_INSTS_MFMA")})
PY
done
