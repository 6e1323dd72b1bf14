#!/bin/bash
# every half second: the processes that hold /dev/kfd open (their command lines), to find who counts against the box's
# process guard.  usage: bash scripts/kfd_holders.sh OUT &   (ends when OUT.stop appears)
out=$1
while [ ! -e $out.stop ]; do
  n=0; names=""
  for p in /proc/[0-9]*; do
    if ls -l $p/fd 2>/dev/null | grep -q "/dev/kfd"; then
      n=$((n+1)); names="$names | $(tr '\0' ' ' < $p/cmdline | cut -c1-90)"
    fi
  done
  echo "$(date +%s.%N | cut -c1-14) holders=$n $names" >> $out
  sleep 0.5
done
