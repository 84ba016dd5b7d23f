"""DoA consumer: the toolkit-free controller of the reference's compass widget
(python/doa_compass_control.py:23-114; plotter contract python/doa_compass_plotter.py:141-199).  CPU only."""
import math

import importlib.util
import os

import pytest

# loaded from its file: the controller is plain python, and importing the `gr_baz_amd.baz` package here would load the
# native module (and with it the HIP runtime) while pytest is still collecting
_spec = importlib.util.spec_from_file_location(
    "doa_compass_control", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gr_baz_amd", "baz",
                                        "doa_compass_control.py"))
dcc = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(dcc)


def test_constants_and_surface_match_the_reference():
    assert (dcc.BEAM_AZM_KEY, dcc.BEAM_ENB_KEY) == ("beam_azm", "beam_enb")
    assert (dcc.POINTER_WIDTH, dcc.SLIDER_STEP_SIZE, dcc.BEAM_COLOR_SPEC, dcc.PLOTTER_SIZE) == (3, 3, (0, 0, 1), (450, 450))
    for name in ("update", "set_direction", "set_text", "set_text_visible", "subscribe", "proxy"):
        assert callable(getattr(dcc.compass_control, name))


def test_pointer_geometry_and_hidden_pointer():
    c = dcc.compass_control(None, direction=40.3, text="DoA", text_visible=True)
    color, fill, profile = c.plotter.profiles["1beam_azm"]
    assert color == (0, 0, 1) and fill is True
    assert profile == [(0, 40.3), (1.0, 40.3 - 1.5), (1.0, 40.3 + 1.5)]             # .py:91-95
    (x0, y0), (x1, y1), (x2, y2) = c.plotter.polygons()["1beam_azm"]
    assert (x0, y0) == (0.0, 0.0)
    assert math.isclose(math.degrees(math.atan2(y1, x1)), 38.8) and math.isclose(math.degrees(math.atan2(y2, x2)), 41.8)
    assert c.plotter.shown_text == "DoA"
    c.set_direction(None)                                                          # .py:100-102: pointer hidden
    assert c[dcc.BEAM_ENB_KEY] is False and c.plotter.profiles["1beam_azm"][2] == [] and c.plotter.polygons() == {}
    c.set_direction(121.7)
    assert c[dcc.BEAM_ENB_KEY] is True and c[dcc.BEAM_AZM_KEY] == 121.7
    assert c.plotter.profiles["1beam_azm"][2][0] == (0, 121.7)


def test_direction_is_proxied_onto_the_flowgraph_pubsub():
    ps = dcc.pubsub()
    seen = []
    c = dcc.compass_control(None, ps=ps, direction_key="doa", callback=seen.append, direction=10.0)
    assert ps["doa"] == 10.0 and c[dcc.BEAM_AZM_KEY] == 10.0 and seen == []      # construction does not fire the callback
    n = c.plotter.updates
    ps["doa"] = 200.0                                                              # the flowgraph side writes
    assert c[dcc.BEAM_AZM_KEY] == 200.0 and c.plotter.profiles["1beam_azm"][2][0] == (0, 200.0)
    assert c.plotter.updates > n and seen == [200.0]
    c.set_direction(90.0)                                                          # the widget side writes
    assert ps["doa"] == 90.0 and seen == [200.0, 90.0]


def test_text_rules_of_the_plotter():
    p = dcc.recording_plotter()
    p.set_text_visible(None, True)
    p.set_text("hidden")                          # visible never set: nothing shows (text_visible False)
    assert p.shown_text is None
    p.set_text_visible(True)
    assert p.shown_text == "hidden"
    p.set_text(["north", "south"])
    assert p.shown_text == "north"                # True = first entry
    p.set_text_visible(1)                         # True == 1 in python: "unchanged", as in the reference (.py:165-168)
    assert p.shown_text == "north"
    p.set_text_visible(1, True)                   # forced: an int indexes the list
    assert p.shown_text == "south"
    p.set_text_visible(-1)                        # negative index: keeps what is shown, draws nothing new
    assert p.shown_text == "south"
    p.set_text("")                                # empty text: no change
    assert p.shown_text == "south"


@pytest.mark.parametrize("ang, lvl, want", [
    ([40.3, 121.7], [9.0, 4.0], 40.3),
    ([40.3, 121.7], [4.0, 9.0], 121.7),
    ([40.3, 40.4], [5.0, 5.0], 40.3),            # tie: the earlier slot (the block's order)
    ([0.0, 0.0], [0.0, 0.0], None),              # unused slots
    ([], [], None),
    ([7.0], None, 7.0),
])
def test_strongest_direction(ang, lvl, want):
    assert dcc.strongest_direction(ang, lvl) == want


def test_compass_follows_the_golden_doa_stream():
    """The consumer wired behind the block's ports 0 / 1 (as a GRC function probe would): every golden item's strongest
    estimate becomes the pointer direction."""
    import numpy as np
    from conftest import load_golden
    g = load_golden("cfg1_m4_n2_N256_r360")
    c = dcc.compass_control(None, text=["searching", "locked"], text_visible=0)
    assert c[dcc.BEAM_ENB_KEY] is False and c.plotter.polygons() == {} and c.plotter.shown_text == "searching"
    for ang, lvl in zip(g["ang"], g["lvl"]):
        d = dcc.strongest_direction(ang, lvl)
        assert d == float(ang[int(np.argmax(lvl))]) and 0.0 <= d < 360.0      # ports 0 / 1 are in descending strength
        c.set_direction(d)
        assert c.plotter.profiles["1beam_azm"][2] == dcc.pointer_profile(d)
    c.set_text_visible(1)
    assert c.plotter.shown_text == "locked"
    c.set_direction(dcc.strongest_direction([0.0, 0.0], [0.0, 0.0]))           # an item without estimates hides the pointer
    assert c.plotter.polygons() == {}
