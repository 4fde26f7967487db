echo "== entry marks (EB_X=13: marks + preload + loads first)"
T="python scripts/trace_rollout.py --lib /tmp/ab/libx1.so"
$T --n-env 65536 2>&1 | grep -E "launch|rec wave|block (start|dur)"
$T --n-env 32768 2>&1 | grep -E "launch|rec wave|block (start|dur)"
echo "-- 1x4 tile (2 waves per block): 8192 envs = 1024 blocks, 16384 envs = 2048 blocks"
$T --n-env 8192 --tile 2 --waves 2 2>&1 | grep -E "launch|rec wave|block (start|dur)"
$T --n-env 16384 --tile 2 --waves 2 2>&1 | grep -E "launch|rec wave|block (start|dur)"
echo "-- 4x4 tile: 32768 envs = 1024 blocks x 5 waves"
$T --n-env 32768 --tile 1 2>&1 | grep -E "launch|rec wave|block (start|dur)"
echo "-- x13"
python scripts/trace_rollout.py --lib /tmp/ab/libx13.so --n-env 65536 2>&1 | grep -E "launch|rec wave|block (start|dur)"
python scripts/trace_rollout.py --lib /tmp/ab/libx13.so --n-env 32768 2>&1 | grep -E "launch|rec wave|block (start|dur)"
for rep in 1 2; do
  for x in base 12 14 10; do
    lib=""; [ $x != base ] && lib="--lib /tmp/ab/libx$x.so"
    for cfg in "4096 16" "16384 32" "32768 32" "65536 32" "131072 32" "262144 32" "65536 64"; do
      set -- $cfg
      echo -n "rep $rep x=$x: "; python scripts/time_rollout.py $lib --n-env $1 --n-veh $2 --iters 2000 2>&1 | grep "us/step  "
    done
  done
done
