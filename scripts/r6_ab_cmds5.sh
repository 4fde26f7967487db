for rep in 1 2 3; do
  for x in base x12 x12p pf3p x12q pf3q x12r pf3r pf4q; do
    lib=""; [ $x != base ] && lib="--lib /tmp/ab/lib$x.so"
    for cfg in "4096 16" "32768 32" "65536 32" "262144 32" "65536 64 --f16"; do
      set -- $cfg
      echo -n "rep $rep x=$x: "; python scripts/time_rollout.py $lib --n-env $1 --n-veh $2 $3 --iters 2000 2>&1 | grep "us/step  "
    done
  done
done
