# sourced by the probe scripts: build ONE source of the library with extra flags and link a variant library in /tmp.
# The shipped peppa_amd/libpeppa_hip.so is never touched; the probes reach the variant through PEPPA_HIP_LIB and, because
# variant / ablation builds report themselves (pp_experimental_build), PEPPA_ALLOW_EXPERIMENTAL=1.
#   variant_lib <source stem, e.g. wgrad_tw> <flags...>   ->  $VARIANT_LIB
VARIANT_DIR=$(mktemp -d /tmp/peppa_variant.XXXXXX)
trap 'rm -rf "$VARIANT_DIR"' EXIT
VARIANT_LIB=$VARIANT_DIR/libpeppa_hip.so
BASE_FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Xclang -target-feature -Xclang -packed-fp32-ops"   # = peppa_amd/build.py FLAGS; NO_BASE_FLAGS=1: plain -O3 (packed FP32 back)
variant_lib() {
  local stem=$1; shift
  local objs=$(ls peppa_amd/build/*.o | grep -v "/$stem.o")
  local flags=$BASE_FLAGS; [ -n "$NO_BASE_FLAGS" ] && flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
  /opt/rocm/bin/hipcc $flags "$@" -c peppa_amd/csrc/$stem.hip -o $VARIANT_DIR/$stem.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $VARIANT_LIB $objs $VARIANT_DIR/$stem.o
  export PEPPA_HIP_LIB=$VARIANT_LIB PEPPA_ALLOW_EXPERIMENTAL=1
}
