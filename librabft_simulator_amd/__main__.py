"""`python -m librabft_simulator_amd` -- the reference's `librabft_simulator` CLI (librabft-v2/src/main.rs:72-172)
on the HIP path, same flag names and defaults, plus `--instances` to run a batch of seeds
(seed, seed+1, ...) in lockstep on one GPU."""
import argparse
import random
import sys

import numpy as np


def get_arguments(argv=None):
    ap = argparse.ArgumentParser(prog="librabft_simulator", description="Simulate LibraBFT v2 (MI355X HIP path)")
    ap.add_argument("--max_clock", type=int, default=1000)
    ap.add_argument("--mean", type=float, default=10.0, help="Mean of the log-normal network delay")
    ap.add_argument("--variance", type=float, default=4.0, help="Variance of the log-normal network delay")
    ap.add_argument("--nodes", type=int, default=3)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--commands_per_epoch", type=int, default=30000)
    ap.add_argument("--target_commit_interval", type=int, default=100000)
    ap.add_argument("--delta", type=int, default=20)
    ap.add_argument("--gamma", type=float, default=2.0)
    ap.add_argument("--lambda", dest="lambda_", type=float, default=0.5)
    ap.add_argument("--create_csv", "--output_data_files", dest="output_data_files", default=None,
                    help="directory for round_switches.txt / number_of_messages.txt (bft-lib/src/data_writer.rs), instance 0")
    ap.add_argument("--instances", type=int, default=1, help="extension: number of independent networks (seeds seed+i)")
    ap.add_argument("--device", type=int, default=0)
    return ap.parse_args(argv)


def main(argv=None):
    args = get_arguments(argv)
    from . import BatchSimulator, NodeConfig, RandomDelay
    seed = args.seed if args.seed is not None else random.getrandbits(64)
    print("seed: %d" % seed, file=sys.stderr)
    seeds = (np.arange(args.instances, dtype=np.uint64) + np.uint64(seed)).astype(np.uint64)
    sim = BatchSimulator.new(seeds, args.nodes, RandomDelay.new(args.mean, args.variance),
                             NodeConfig(args.target_commit_interval, args.delta, args.gamma, args.lambda_),
                             commands_per_epoch=args.commands_per_epoch, device=args.device)
    res = sim.loop_until(args.max_clock, args.output_data_files)
    cc = res.commit_counts
    if args.instances == 1:
        # main.rs:47-53
        print("Commands executed per node: %s" % [int(v) for v in cc[0]])
    else:
        c = res.counters
        print("Commands executed per node (instance 0): %s" % [int(v) for v in cc[0]])
        print("instances %d: rounds %d, commits %d, events %d" % (args.instances, c["rounds"], c["commits"], sum(c["events"])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
