#!/bin/bash
# register / scratch / occupancy of the kernels of one build unit whose mangled name matches a pattern:
#   tools/kres.sh <geom 0-5> <part 0-4> <name pattern> [extra hipcc flags...]
G=$1; PART=$2; PAT=$3; shift 3
cd "$(dirname "$0")/../hyperion_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -fPIC -DHYP_GEOM_TU=$G -DHYP_PART=$PART "$@" -c hyp_geom.hip -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk -v pat="$PAT" '/Function Name/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-R.*/,"",name); show=(name ~ pat)} show && /VGPRs:|ScratchSize|Occupancy|VGPRs Spill|LDS Size/ {v=$0; sub(/.*remark: +/,"",v); sub(/ \[-R.*/,"",v); line=line " | " v} /LDS Size/ && show {print name line; line=""}'
rm -f /tmp/kres_$$.o
