#!/bin/bash
# registers / spills / scratch of every non-counting trace_shade_kernel instantiation: tools/kres.sh [extra hipcc flags]
cd "$(dirname "$0")/../ray-tracer_amd" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 "$@" \
  -Rpass-analysis=kernel-resource-usage -c csrc/rtx_kernel.hip -o /tmp/kres.o 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"Function Name: _Z18trace_shade_kernelILi([012])ELb([01])ELb([01])ELb([01])ELb([01])", line)
    if m:
        cur = "CSG=%s COUNT=%s PROBE=%s SPLIT=%s HCTX=%s" % m.groups(); vals = []
        continue
    m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        vals.append("%s %s" % (m.group(1).split(" [")[0], m.group(2)))
        if len(vals) == 4 and "COUNT=0" in cur:
            print(cur, " | ".join(vals))
'
