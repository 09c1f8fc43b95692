# thresholds of the deferred propagation kernel with the forced-first pre-pass (lanes that wait for an interaction / an emission before the
# wave runs that code), configs[3] imaging at 1e8 packets
for it in ${ITS:-8 16 24 32}; do for et in ${ETS:-16 32 48}; do
  echo -n "interact $it emit $et: "; python tools/r03_workload.py oct_img 1e8 final_interact_threshold=$it final_emit_threshold=$et 2>&1 | grep "^oct_img" | tail -1 | cut -c1-60
done; done
