#!/bin/bash
# compact per-kernel resource table of one HIP source (no GPU needed):  tools/kres.sh rapiddoc_amd/csrc/kernels_mixer_ws.hip [filter]
SRC=$1; FIL=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c "$SRC" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)} /VGPRs:/ {v=$NF; sub(/.*VGPRs: /,"",$0); v=$1} / VGPRs Spill:/ {s=$0; sub(/.*Spill: /,"",s); sub(/ .*/,"",s)} /ScratchSize/ {c=$0; sub(/.*: /,"",c); sub(/ .*/,"",c)} /TotalSGPRs/ {g=$0; sub(/.*: /,"",g); sub(/ .*/,"",g)} /LDS Size/ {print n, "vgpr="v, "spill="s, "scratch="c, "sgpr="g}' \
 | grep -E "$FIL" | while read n rest; do echo "$(echo $n | c++filt | sed "s/(.*//" | cut -c1-80) $rest"; done
