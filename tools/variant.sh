# Sourced by the measurement scripts: development builds of the C-ABI library live under build/variants/ and are picked
# up through GI2D_LIB; the product library in the tree is never rebuilt with development switches (csrc/Makefile).
#   use_variant "<EXTRA flags>"   builds (if needed) and selects the variant for every python started afterwards
#   use_product                   back to gaussianimage_plus_amd/libgi2d_hip.so
: ${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
variant_name() {  # a file-system-safe name for a flag string
  local n; n=$(echo "$1" | sed 's/-DGI2D_//g' | tr -c 'A-Za-z0-9=\n' '_' | sed 's/^_*//; s/_*$//')
  echo "${n:-plain}"
}
use_variant() {
  local flags="$1" name; name=$(variant_name "$1")
  if [ -z "$(echo $flags)" ]; then use_product; return; fi
  # PREBUILT=1: variants were built before the snapshot was sent (file times do not survive the copy: make would rebuild)
  if [ "$PREBUILT" = 1 ] && [ -f $GRAFT_REPO_ROOT/build/variants/$name/libgi2d_hip.so ]; then
    export GI2D_LIB=$GRAFT_REPO_ROOT/build/variants/$name/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1; return
  fi
  if make -s -j8 -C $GRAFT_REPO_ROOT/gaussianimage_plus_amd/csrc VARIANT="$name" EXTRA="$flags" 2>&1 | grep -E "error|Error"; then return 1; fi
  export GI2D_LIB=$GRAFT_REPO_ROOT/build/variants/$name/libgi2d_hip.so GI2D_ALLOW_DEV_BUILD=1
}
use_product() { unset GI2D_LIB GI2D_ALLOW_DEV_BUILD; }
