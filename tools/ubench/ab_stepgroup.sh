#!/bin/bash
# stand-alone decoder-step launch group (tools/step_group_run.py) under several builds / diagnosis switches on one box.
# usage: ab_stepgroup.sh "<lib.so> [ENV=val ...]" ...   (lib relative to controllable_xgating_amd/lib; 3 rounds, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
cfgs=("$@")
for rep in 1 2 3; do for cfg in "${cfgs[@]}"; do
  read -r -a w <<< "$cfg"; lib=${w[0]}; envs=("${w[@]:1}")
  printf "%-28s %-28s " "$lib" "${envs[*]}"
  env "${envs[@]}" XG_LIBRARY=$PWD/controllable_xgating_amd/lib/$lib timeout 200 python tools/step_group_run.py 300 2>&1 | tail -1
done; done
