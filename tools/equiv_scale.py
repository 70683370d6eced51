#!/usr/bin/env python3
"""Runs on the GPU box: the headline batch shape with one equivocating leader per network (kernel class 1, lean kernel
lbft_k_run1l), with the automatic queue discipline and with the calendar switched off."""
import sys, json
sys.path.insert(0, '.')
import numpy as np
from librabft_simulator_amd import BatchSimulator, RandomDelay, NodeConfig
for lpw, cal in ((0, True), (0, False)):
    seeds = np.arange(1, 65537, dtype=np.uint64)
    sim = BatchSimulator.new(seeds, 4, RandomDelay.new(10.0, 4.0), NodeConfig(), equivocate_every=4, lanes_per_wavefront=lpw, calendar_queue=cal)
    ms = []
    for _ in range(3):
        sim.reset(); res = sim.loop_until(1000); ms.append(sim.last_run_ms()[1])
    print("65536x4 one equivocator per network: calendar", cal, "lpw", sim.layout()["lanes_per_wavefront"], "class", sim.layout()["kernel_class"], "ms", round(min(ms), 2), "events", sum(res.counters["events"]), "faulted", res.counters["faulted_instances"])
