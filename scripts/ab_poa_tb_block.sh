# poa's team traceback with the register block serving second / third in-edge sources too (in tree) against first sources only
# (build_tmp/libgbx_poa_tb_r05.so = -DGBX_POA_TB_BLOCK23=0): lone windows, a config-4 shard, and the 8-shard prediction
for lib in "" build_tmp/libgbx_poa_tb_r05.so "" build_tmp/libgbx_poa_tb_r05.so; do
  echo "== lib ${lib:-intree}"
  GBX_LIB=${lib:+$PWD/$lib} python scripts/dbg_poa_lone.py 64 750 2>/dev/null | grep windows:
done
for lib in "" build_tmp/libgbx_poa_tb_r05.so; do
  echo "== predict 8 shards, lib ${lib:-intree}"
  GBX_LIB=${lib:+$PWD/$lib} python bench.py --kernel poa --predict-shards 8 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config4_predicted']
print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('predicted_ms_per_step','predicted_speedup','whole_job_ms_1gpu','verified')})"
done
