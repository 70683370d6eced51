"""The round-switch plotter (counterpart of bft-lib/src/visualization/round_switch/round_plotter.py) on a file in the DataWriter layout."""
import os

import pytest

from librabft_simulator_amd.simulator import write_data_files
from librabft_simulator_amd.visualization import round_plotter


def test_step_curves_follow_the_first_seen_times(tmp_path):
    rows = [[0, 0, 0], [10, 11, 8], [35, None, 33], [61, 64, 60]]  # round r first seen at these times; node 1 skipped round 2
    write_data_files(str(tmp_path), rows, 42, 3)
    header, table = round_plotter.read_round_switches(os.path.join(str(tmp_path), "round_switches.txt"))
    assert header == ["node 0", "node 1", "node 2"] and table == rows
    curves = round_plotter.step_curves(table, 3, tail=100)
    assert curves[0] == ([0, 0, 10, 35, 61, 164], [0, 0, 1, 2, 3, 3])
    assert curves[1] == ([0, 0, 11, 64, 164], [0, 0, 1, 3, 3])          # straight from round 1 to round 3
    assert all(ts[-1] == 164 for ts, _ in curves)                        # same length for every node (the reference pads likewise)


def test_plot_writes_a_figure(tmp_path):
    pytest.importorskip("matplotlib")
    write_data_files(str(tmp_path), [[0, 0], [9, 12], [30, 31]], 7, 2)
    out = os.path.join(str(tmp_path), "rounds.png")
    round_plotter.main([os.path.join(str(tmp_path), "round_switches.txt"), "-o", out, "--no-show"])
    assert os.path.getsize(out) > 1000
