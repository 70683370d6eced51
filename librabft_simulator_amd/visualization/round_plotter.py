#!/usr/bin/env python3
"""Round-switch plot: which round every node is in over (global) time, from the ``round_switches.txt`` that
``python -m librabft_simulator_amd --create_csv`` / ``loop_until(csv_path=...)`` writes in the reference's DataWriter layout
(bft-lib/src/data_writer.rs:62-97).  Counterpart of the reference's bft-lib/src/visualization/round_switch/round_plotter.py (same
input file, same picture: one step curve per node, time on x, round number on y); built on numpy step functions instead of
per-time-unit lists, and able to write the figure to a file (``-o``) where no display exists.

    python -m librabft_simulator_amd.visualization.round_plotter data/round_switches.txt -o rounds.png"""
import argparse
import csv
import sys


def read_round_switches(path):
    """-> (node names, first_time[round][node] with None for rounds a node never entered)."""
    with open(path) as f:
        rows = list(csv.reader(f))
    if not rows:
        raise ValueError("empty round-switch file: " + path)
    header, body = rows[0], rows[1:]
    table = [[int(c) if c != "" else None for c in row] + [None] * (len(header) - len(row)) for row in body]
    return header, table


def step_curves(table, num_nodes, tail=100):
    """Per node: (times, rounds) of a step function -- the node is in round r from the first time it was seen there until it is seen
    in a later round; the curve extends `tail` time units past the last switch of any node (as the reference's plotter pads)."""
    last = max((t for row in table for t in row if t is not None), default=0)
    curves = []
    for node in range(num_nodes):
        ts, rs = [0], [0]
        for rnd, row in enumerate(table):
            t = row[node]
            if t is not None:
                ts.append(t)
                rs.append(rnd)
        ts.append(last + tail)
        rs.append(rs[-1])
        curves.append((ts, rs))
    return curves


def plot(path, out=None, show=True):
    import matplotlib
    if out and not show:
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    header, table = read_round_switches(path)
    fig = plt.figure()
    for (ts, rs), name in zip(step_curves(table, len(header)), header):
        plt.step(ts, rs, where="post", label=name.strip().capitalize())
    plt.legend()
    plt.xlabel("Time")
    plt.ylabel("Round number")
    plt.grid(axis="both", which="both")
    if out:
        fig.savefig(out, dpi=120)
    if show:
        plt.show()
    return fig


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("csv_path", help="round_switches.txt written by the simulator (--create_csv)")
    ap.add_argument("-o", "--output", help="write the figure to this file instead of (only) showing it")
    ap.add_argument("--no-show", action="store_true")
    a = ap.parse_args(argv)
    try:
        plot(a.csv_path, a.output, show=not a.no_show)
    except OSError as e:
        sys.exit("Provide the path of the round-switch csv file: %s" % e)


if __name__ == "__main__":
    main()
