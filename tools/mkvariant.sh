#!/bin/bash
# Build a diagnostic variant of the library: tools/mkvariant.sh <prefix_NAME> <flags...>  ->  new-vit_amd/mst/hip/lib<prefix_NAME>.so
# (own object directory; the default library is relinked from its own objects afterwards).  Variants are git-ignored and travel to the GPU box.
set -e
NAME=$1; shift
MST_EXTRA_FLAGS="$*" python new-vit_amd/build.py > /dev/null
cp new-vit_amd/mst/hip/libmst_hip.so new-vit_amd/mst/hip/lib$NAME.so
rm -f new-vit_amd/mst/hip/libmst_hip.so
python new-vit_amd/build.py | tail -1
