#!/bin/bash
# A/B of several builds of the library on one box: alternate processes (HPC_RLL_LIB selects the .so), several rounds.
# usage: LIBS="new tests/tools/micro/libhpc_old.so ..." lstm_lib_ab.sh "<S B I H L>" ...     ("new" = the shipped library)
for sh in "$@"; do
  for r in 1 2; do
    for lib in $LIBS; do
      if [ $lib = new ]; then unset HPC_RLL_LIB; else export HPC_RLL_LIB=$lib; fi
      echo -n "$(basename $lib): "; timeout 120 python tests/tools/lstm_small_probe.py $sh 2>&1 | grep "persist=1" | tail -1
    done
  done
done
