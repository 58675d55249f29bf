"""The reference's CONSUMERS against this repository's PlotEngine ("GUI.py and ExportManager.py drop in
unchanged").  tests/golden/g6_consumers.npz holds what the reference's own PlotEngine + ExportManager produced in the build
container for GUI.plot_selected's call sequence (GUI.py:374-453), Auto-Detect (GUI.py:455-476), Learn (GUI.py:286-312) and
the burst CSV export (ExportManager.py:13-90); here the same sequence runs on the device-backed engine and must give the
same engine state, the same events and the same CSV text.  hmmlearn is absent on both sides: tests/hmm_standin.py is the
model for both (the logic around it is what is compared).  The exporter below restates export_to_csv's attribute reads."""
import csv
import io
import os
import re
import sys
import warnings

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

SETTINGS = {"combine": False, "draw_raw": True, "draw_proc": False, "mode_raw": "Both", "mode_proc": "None",
            "nperseg": 256, "fmin": 5.0, "fmax": 30.0, "log_scale": False}
USER_ROLE = 256                                              # QtCore.Qt.UserRole


class FakeItem:
    def __init__(self, name):
        self.name = name

    def data(self, column, role):
        assert column == 0 and role == USER_ROLE             # the only read the reference ever makes (GUI.py:388)
        return self.name


def consumer_sweeps():                                       # same recipe as tests/golden/make_golden.py
    rng = np.random.default_rng(11)
    out = []
    for i, (name, n) in enumerate((("/data/recA_sweep0", 6000), ("/data/recA_sweep1", 5000), ("/data/recB_sweep3", 7000))):
        t = np.arange(n) / 500.0
        x = 0.2 * rng.standard_normal(n) + 1.0
        for b0 in (2.0 + i, 7.0 + 0.5 * i):
            x += np.where((t > b0) & (t < b0 + 1.5), 2.0 * np.sin(2 * np.pi * 12 * t), 0.0)
        out.append((name, x))
    return out


def export_csv_text(engine):
    """What ExportManager.export_to_csv (ExportManager.py:13-90) reads from the engine and writes: returns (message, text)."""
    if not engine.burst_patches:
        return "Error: No burst data to export.", ""
    segment_map, items = engine.segment_map, engine.currently_plotted_items
    if hasattr(engine, "last_detected_events"):
        pairs = list(engine.last_detected_events)
    else:
        pairs = [(min(p.get_extents().x0, p.get_extents().x1), max(p.get_extents().x0, p.get_extents().x1)) for p, _ in engine.burst_patches]
    pairs = sorted(pairs)

    def source_of(name):
        m = re.search(r"_sweep(\d+)$", name)
        return re.sub(r"_sweep\d+$", "", os.path.basename(name)), (m.group(1) if m else "Unknown")

    rows = []
    for i, (t0, t1) in enumerate(pairs):
        gap = np.nan if i == 0 else t0 - pairs[i - 1][1]
        src, sweep = "Unknown", "Unknown"
        if segment_map:
            for seg in segment_map:
                if seg["start_time_combined"] <= t0 < seg["end_time_combined"]:
                    src, sweep = source_of(seg["source_item"].data(0, USER_ROLE))
                    break
        elif items:
            src, sweep = source_of(items[0].data(0, USER_ROLE))
        rows.append([i + 1, src, sweep, t0, t1, gap])
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    w.writerow(["Burst ID", "Source File", "Sweep", "Start Time (s)", "End Time (s)", "Inter Burst Interval (s)"])
    w.writerows(rows)
    return f"Successfully exported {len(rows)} events to bursts.csv", buf.getvalue().replace("\r\n", "\n")


@pytest.fixture()
def engine_cls():
    import hmm_standin
    saved = {k: sys.modules.get(k) for k in ("hmmlearn", "hmmlearn.hmm")}
    hmm_standin.install(sys.modules)
    from PlotEngine import PlotEngine
    yield PlotEngine
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


@pytest.mark.parametrize("tag", ["single", "combined"])
def test_gui_and_export_sequence_matches_reference(engine_cls, tag):
    g = load_golden("g6_consumers.npz")
    combine = tag == "combined"
    sweeps = consumer_sweeps()
    eng = engine_cls(parent=None)                              # GUI.py:157
    infos = [{"item": FakeItem(n), "signal_raw": x, "signal_proc": None, "fs": 500.0} for n, x in (sweeps if combine else sweeps[:1])]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng.set_editing_enabled(False)                         # GUI.py:434
        eng.plot_sweeps(infos, dict(SETTINGS, combine=combine))    # GUI.py:437
        eng.draw()                                             # GUI.py:448
    np.testing.assert_array_equal(eng.last_t, g[f"{tag}__last_t"])
    np.testing.assert_array_equal(eng.last_f, g[f"{tag}__last_f"])
    assert np.isclose(eng.calculate_absolute_power(), float(g[f"{tag}__abs_power"]), rtol=1e-10)      # GUI.py:451 (f64 signal)
    segs = np.array([[s["start_time_combined"], s["end_time_combined"]] for s in eng.segment_map]).reshape(-1, 2)
    np.testing.assert_array_equal(segs, g[f"{tag}__segments"])
    if combine:
        assert [s["source_item"].name for s in eng.segment_map] == [str(n) for n in g[f"{tag}__segment_names"]]
        assert eng.last_raw_t[-1] == float(g[f"{tag}__last_raw_t_end"])
        assert len(eng.combined_raw) == int(g[f"{tag}__combined_raw_len"])
    else:
        assert len(eng.last_raw_t) == 0 and eng.combined_raw is None
    assert export_csv_text(eng)[0] == str(g[f"{tag}__export_empty_msg"])

    # ---- Auto-Detect (GUI.py:455-476)
    ev = eng.unsupervised_detect()
    np.testing.assert_array_equal(np.array(ev, dtype=np.float64).reshape(-1, 2), g[f"{tag}__auto_events"])
    assert np.allclose(eng.model.transmat_, g[f"{tag}__auto_transmat"], rtol=0, atol=1e-12)      # incl. the escape-route surgery
    assert eng.last_detected_events == ev
    eng.plot_detection_lines(ev)
    assert len(eng.burst_patches) == int(g[f"{tag}__auto_n_patches"])
    msg, text = export_csv_text(eng)
    assert msg == str(g[f"{tag}__auto_csv_msg"]) and text == str(g[f"{tag}__auto_csv"])

    # ---- Learn (GUI.py:286-312): regions "drawn" around known bursts
    eng.plot_detection_lines([tuple(r) for r in g[f"{tag}__learn_rois"]])
    ev2 = eng.learn_and_detect()
    np.testing.assert_array_equal(np.array(ev2, dtype=np.float64).reshape(-1, 2), g[f"{tag}__learn_events"])
    assert eng.is_model_refined
    assert np.allclose(eng.model.transmat_, g[f"{tag}__learn_transmat"], rtol=0, atol=1e-12)
    assert np.allclose(eng.model.means_, g[f"{tag}__learn_means"], rtol=0, atol=1e-9)
    eng.plot_detection_lines(ev2)
    msg, text = export_csv_text(eng)
    assert msg == str(g[f"{tag}__learn_csv_msg"]) and text == str(g[f"{tag}__learn_csv"])
    # refined model: Auto-Detect decodes without refitting (PlotEngine.py:417)
    ev3 = eng.unsupervised_detect()
    np.testing.assert_array_equal(np.array(ev3, dtype=np.float64).reshape(-1, 2), g[f"{tag}__refined_events"])
    eng.reset_model()                                          # GUI.py:352
    assert not eng.is_model_refined


def test_roi_mouse_editor_draw_hover_delete_merge(engine_cls):
    """The mouse editor (PlotEngine.py:480-667) through synthetic matplotlib events: drag to add a region, hover highlights,
    a too-narrow drag adds nothing, Delete / Merge act on the highlighted region and keep last_detected_events in step."""
    from matplotlib.backend_bases import MouseEvent
    sweeps = consumer_sweeps()
    eng = engine_cls()
    info = [{"item": FakeItem(sweeps[0][0]), "signal_raw": sweeps[0][1], "signal_proc": None, "fs": 500.0}]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        eng.plot_sweeps(info, dict(SETTINGS))
        eng.draw()
    eng.set_editing_enabled(True)
    assert eng.press_cid and eng.release_cid and eng.motion_cid

    def fire(kind, ax, x, button=None):
        px, py = ax.transData.transform((x, sum(ax.get_ylim()) / 2))
        ev = MouseEvent(kind, eng, px, py, button=button)
        eng.callbacks.process(kind, ev)

    def drag(ax, x0, x1):
        fire("button_press_event", ax, x0, button=1)
        fire("motion_notify_event", ax, (x0 + x1) / 2)
        assert eng.adding_patch is not None                   # rubber band while dragging
        fire("button_release_event", ax, x1, button=1)
        assert eng.adding_patch is None and not eng.is_adding

    drag(eng.ax_signal, 4.0, 2.0)                             # right-to-left: stored ordered
    drag(eng.ax_spec, 2.5, 3.0)
    drag(eng.ax_spec, 3.2, 3.6)
    drag(eng.ax_signal, 8.0, 8.0001)                          # narrower than one sample (no last_raw_t: 1/fs = 2 ms)
    assert len(eng.burst_patches) == 3 and len(eng.last_detected_events) == 3
    assert np.allclose(eng.burst_patches[0][0].event_data, (2.0, 4.0), atol=0.02)
    inner1, inner2 = eng.burst_patches[1][0].event_data, eng.burst_patches[2][0].event_data
    # hover highlights the first region under the pointer; leaving the axes clears it
    fire("motion_notify_event", eng.ax_signal, 2.2)
    assert eng.hovered_patch is eng.burst_patches[0]
    assert eng.hovered_patch[0].get_facecolor()[:3] == (1.0, 0.0, 0.0)
    # Merge: the two regions inside the hovered one collapse into one event, the container goes
    eng.merge_into_hovered()
    assert len(eng.burst_patches) == 1 and eng.hovered_patch is None
    merged = eng.burst_patches[0][0].event_data
    assert merged == (min(inner1[0], inner2[0]), max(inner1[1], inner2[1]))
    assert eng.last_detected_events == [merged]
    # Delete the remaining one
    fire("motion_notify_event", eng.ax_spec, (merged[0] + merged[1]) / 2)
    assert eng.hovered_patch is eng.burst_patches[0]
    eng.delete_hovered()
    assert eng.burst_patches == [] and eng.hovered_patch is None
    eng.set_editing_enabled(False)
    assert eng.press_cid is None and not eng.editing_enabled
