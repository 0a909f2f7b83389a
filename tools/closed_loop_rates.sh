#!/bin/bash
# config 5 at a real-time cadence: apps/mppi_closed_loop paced at R control steps per second of wall
# time (the reference's MuJoCo loop runs in real time), the next solve's noise drawn ahead
# (MPPI_PREFETCH=1, the default) or not
for cfg in "3 100000 200" "2 10000 200"; do set -- $cfg
  for rate in 0 100 1000; do for pf in 0 1; do
    printf "dims %s K %s T %s  rate %5s Hz  MPPI_PREFETCH=%s  " $1 $2 $3 $rate $pf
    MPPI_PREFETCH=$pf timeout -k 10 120 apps/mppi_closed_loop --dims $1 --samples $2 --horizon $3 --seconds 4 --rate-hz $rate | grep RESULT
  done; done
done
