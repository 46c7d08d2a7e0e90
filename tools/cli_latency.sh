#!/bin/bash
# wall clock of the CLI (scene file -> PNG, like `cargo run --release scene.yaml`) on 4K / 8K scenes: tools/cli_latency.sh
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && python3 -c "
import sys; sys.path[:0]=['tools']
import scenes
open('gpurun_out/s5_8k.yaml','w').write(scenes.make('S5', width=7680, height=4320))
open('gpurun_out/s4_4k.yaml','w').write(scenes.make('S4', width=3840, height=2160))
open('gpurun_out/s3_4k.yaml','w').write(scenes.make('S3', width=3840, height=2160))
" && cd gpurun_out && for f in s5_8k s4_4k s3_4k; do
  for k in 1 2; do
    t0=$(date +%s.%N); timeout -k 10 120 ../ray-tracer_amd/rtx $f.yaml || exit 1; t1=$(date +%s.%N)
    echo "$f run $k: $(python3 -c "print('%.0f ms' % (($t1 - $t0) * 1e3))") wall, png $(stat -c %s $f.png) bytes"
  done
  rm -f $f.png
done
rm -f s5_8k.yaml s4_4k.yaml s3_4k.yaml
