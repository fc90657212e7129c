#!/bin/bash
# Build ablation / experiment variants of the library: tools/exp_build.sh NAME "-DFLAG ..." [NAME "-DFLAG" ...]
# -> build_exp/libnc_NAME.so (timed by tools/exp_trunk.py on the GPU box). Experiment infrastructure only.
set -e
cd "$(dirname "$0")/../nanocaller_amd/csrc"
make -s >/dev/null 2>&1 || make
SRC=${SRC:-nc_cnn}          # which source the flags apply to (SRC=nc_wire tools/exp_build.sh ...)
OBJS=""
for o in nc_ctx nc_scan nc_featurize nc_cnn nc_indel nc_msa nc_wire nc_pipe nc_synth nc_inflate nc_ingest; do [ $o = $SRC ] || OBJS="$OBJS $o.o"; done
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $SRC.hip -o ../../build_exp/${SRC}_$name.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_exp/libnc_$name.so $OBJS ../../build_exp/${SRC}_$name.o nc_bam.o nc_vcf.o nc_align.o -lz -lpthread -ldl
  rm -f ../../build_exp/${SRC}_$name.o
  echo built $name
done
