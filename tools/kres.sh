#!/bin/bash
# usage: tools/kres.sh dms_amd/csrc/sru.hip  -> one line per kernel: name VGPRs SGPRs scratch occupancy LDS
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n,v; n=$3; v=""; next}{v=v" | "$0}END{print n,v}' | while read -r name rest; do echo "$(echo $name | c++filt | cut -c1-90) $rest"; done
