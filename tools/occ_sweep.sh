for l in librtx_w3.so librtx_hip.so librtx_w5.so librtx_w8.so; do echo $l; RTX_LIB=$l timeout -k 10 200 python3 tools/diag.py > gpurun_out/diag_$l.jsonl 2>&1 || break; python3 -c "
import json,sys
for l in open('gpurun_out/diag_$l.jsonl'):
    d=json.loads(l); print(' ', d['workload'], round(d['fast_ms'],3))
"; done
