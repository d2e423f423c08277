#!/bin/bash
# Build a library variant quickly: start from the product's object files, re-compile only the named objects with extra defines.
# usage: bash tools/build_variant.sh <tag> "<DN_DEFINES>" <object-name-pattern> [...]     e.g. rows3 "DN_GEN_MINW=3" dn_generic_nt64
# -> build_variants/lib_<tag>.so (load it with DN_LIB_PATH)
set -e
tag=$1; defs=$2; shift 2
src=degnorm_amd/csrc
rm -rf $src/obj_$tag
cp -a $src/obj $src/obj_$tag
for pat in "$@"; do rm -f $src/obj_$tag/$pat*.o; done
DN_BUILD_TAG=$tag DN_DEFINES="$defs" python -m degnorm_amd.build > /tmp/build_$tag.log 2>&1 || { tail -30 /tmp/build_$tag.log; exit 1; }
ls -la build_variants/lib_$tag.so
