echo "== trace pf4pe (EB_X=237)"
python scripts/trace_rollout.py --lib /tmp/ab/libpf4pet.so --n-env 65536 2>&1 | grep -E "launch|rec wave|env wave|block (start|dur)"
python scripts/trace_rollout.py --lib /tmp/ab/libpf4pet.so --n-env 32768 2>&1 | grep -E "launch|rec wave|env wave|block (start|dur)"
for rep in 1 2 3; do
  for x in base x12 pf3 pf4 pf4p pf3p pf4e pf4pe pf3pe x12e x12p; do
    lib=""; [ $x != base ] && lib="--lib /tmp/ab/lib$x.so"
    for cfg in "16384 32" "32768 32" "65536 32" "262144 32"; do
      set -- $cfg
      echo -n "rep $rep x=$x: "; python scripts/time_rollout.py $lib --n-env $1 --n-veh $2 --iters 2000 2>&1 | grep "us/step  "
    done
  done
done
