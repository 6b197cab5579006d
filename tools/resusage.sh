#!/usr/bin/env bash
# usage: tools/resusage.sh file.hip [extra flags]  -> VGPRs, AGPRs, scratch, LDS, occupancy, kernel name
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -c "$f" -o /tmp/resusage.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 \
 | grep -E "error|Function Name|VGPRs:|AGPRs:|ScratchSize|LDS Size|Occupancy" \
 | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//' \
 | awk '/Function Name/{if(name)print line" "name; name=$3; line=""} /^ *VGPRs:/{line=line" v="$2} /AGPRs/{line=line" a="$2} /Scratch/{line=line" scratch="$3} /LDS/{line=line" lds="$4} /Occupancy/{line=line" occ="$3} /error/{print} END{print line" "name}' | c++filt | cut -c1-110
