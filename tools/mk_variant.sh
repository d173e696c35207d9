#!/bin/bash
# (local) build the library of git revision $1 into lewton_amd/_lib/variant_$2.so ; "WORK" = the working tree
REV=$1; NAME=$2
set -e
if [ "$REV" = "WORK" ]; then
  python lewton_amd/build.py --force > /dev/null
  cp lewton_amd/_lib/liblewton_amd.so lewton_amd/_lib/variant_$NAME.so
else
  T=$(mktemp -d)
  git archive $REV | tar -x -C $T
  (cd $T && python lewton_amd/build.py --force > /dev/null)
  cp $T/lewton_amd/_lib/liblewton_amd.so lewton_amd/_lib/variant_$NAME.so
  rm -rf $T
fi
echo built variant_$NAME from $REV
