"""DoA consumer of the MUSIC path: the controller behind GRC's "WX DOA Compass" variable block,
without the widget toolkit.

Reference: /root/reference/python/doa_compass_control.py:23-114 (`compass_control`: keys, pointer geometry,
`set_direction` / `set_text` / `set_text_visible`), python/doa_compass_plotter.py:21-22,141-199 (what a plotter stores:
profiles keyed by name, the text and which of its entries shows), grc/doa_compass.xml (how a flowgraph constructs and
retunes it: `direction=`, `text=`, `text_visible=` and the three callbacks).

The reference class is a wx.Panel that owns an OpenGL `compass_plotter`; wxPython, PyOpenGL and gnuradio.wxgui are a GUI
stack this repository does not ship.  What a flowgraph *does* with the widget is state: a direction in degrees (or
None = pointer hidden), a label, and the pointer polygon handed to the plotter.  That state machine is here, with the
drawing surface passed in:

    compass_control(parent, ps=None, direction_key=..., callback=None, direction=None, text=None, text_visible=None,
                    plotter=None)

`plotter` is any object with `set_profile(key=, color_spec=, fill=, profile=)`, `set_text(text, visible=None)`,
`set_text_visible(visible, force=False)` and `update()` -- the reference's `compass_plotter` qualifies; by default a
`recording_plotter` keeps what it was given (and converts profiles to the plotter's rectangular coordinates), which is
what the tests and a headless flowgraph use.  `parent` is accepted and ignored unless the plotter factory wants it.

Differences from the reference, on purpose: `callback` is subscribed to direction changes (the reference subscribes it
to an undefined `TAPS_KEY`, a NameError as soon as a callback is passed, .py:72); the dead `update_enables`
(.py:74-84, undefined `PATTERNS`) is not carried over.

`strongest_direction(ang, lvl)` is the usual glue between `baz.music_doa`'s first two ports and the compass: the angle
of the strongest of an item's n estimates (the block already emits them in descending strength, lib/baz_music_doa.cc:129-141,
so it is `ang[0]` unless `lvl[0]` is 0 = "no estimate").
"""
import math

BEAM_AZM_KEY = 'beam_azm'
BEAM_ENB_KEY = 'beam_enb'

POINTER_WIDTH = 3       # degrees
SLIDER_STEP_SIZE = 3    # degrees
BEAM_COLOR_SPEC = (0, 0, 1)
PLOTTER_SIZE = (450, 450)


class pubsub(dict):
    """The slice of `gnuradio.gr.pubsub` the controller uses: a dict whose writes notify subscribers, and keys that can
    be proxied onto another pubsub's key (reads and writes go there, its subscribers fire)."""

    def __init__(self):
        dict.__init__(self)
        self._subscribers = {}
        self._proxies = {}

    def __missing__(self, key):
        dict.__setitem__(self, key, None)
        return None

    def __setitem__(self, key, val):
        if key in self._proxies:
            other, other_key = self._proxies[key]
            other[other_key] = val
        else:
            dict.__setitem__(self, key, val)
        for fn in list(self._subscribers.get(key, ())):
            fn(val)

    def __getitem__(self, key):
        if key in self._proxies:
            other, other_key = self._proxies[key]
            return other[other_key]
        return dict.__getitem__(self, key)

    def subscribe(self, key, subscriber):
        self._subscribers.setdefault(key, []).append(subscriber)

    def unsubscribe(self, key, subscriber):
        self._subscribers.get(key, []).remove(subscriber)

    def proxy(self, key, other, other_key):
        self._proxies[key] = (other, other_key)

    def unproxy(self, key):
        self._proxies.pop(key, None)


def polar2rect(*coors):
    """(radius, angle in degrees) pairs -> (x, y) pairs, as the plotter draws them (doa_compass_plotter.py:21-22)."""
    return [(r * math.cos(math.radians(a)), r * math.sin(math.radians(a))) for r, a in coors]


class recording_plotter(object):
    """Drawing surface stand-in: keeps the last profile per key, the text and the visible entry, counts updates."""

    def __init__(self, parent=None):
        self.parent = parent
        self.profiles = {}
        self.text = None
        self.text_visible = False
        self.shown_text = None
        self.updates = 0

    def set_profile(self, key='', color_spec=(0, 0, 0), fill=True, profile=()):
        self.profiles[key] = (tuple(color_spec), bool(fill), list(profile))

    def polygons(self):
        """key -> vertices in the plotter's rectangular frame; hidden (empty) profiles are left out."""
        return dict((k, polar2rect(*p)) for k, (_, _, p) in sorted(self.profiles.items()) if p)

    def set_text(self, text, visible=None):
        if self.text == text:
            return
        self.text = text
        if visible is not None:
            self.text_visible = visible
        self._update_text()

    def set_text_visible(self, visible, force=False):
        if not force and self.text_visible == visible:
            return
        self.text_visible = visible
        self._update_text()

    def _update_text(self):
        # which string shows (doa_compass_plotter.py:174-199): nothing to do for an empty text, False or a negative
        # index; True = first entry; an int indexes a list of texts
        if self.text is None or len(self.text) == 0:
            return
        vis = self.text_visible
        if vis is None or (isinstance(vis, bool) and not vis):
            return
        if isinstance(vis, int) and not isinstance(vis, bool) and vis < 0:
            return
        idx = 0 if isinstance(vis, bool) else vis
        self.shown_text = self.text[idx] if isinstance(self.text, list) else self.text

    def update(self):
        self.updates += 1


def pointer_profile(azimuth, width=POINTER_WIDTH):
    """The beam pointer: a sliver from the centre to the rim, `width` degrees wide at the rim (.py:91-95)."""
    return [(0, azimuth), (1.0, azimuth - width / 2.0), (1.0, azimuth + width / 2.0)]


class compass_control(pubsub):
    def __init__(self, parent=None, ps=None, direction_key='__direction_key__', callback=None, direction=None, text=None,
                 text_visible=None, plotter=None):
        if ps is None:
            ps = pubsub()
        if direction is not None:
            ps[direction_key] = direction
        pubsub.__init__(self)
        self.proxy(BEAM_AZM_KEY, ps, direction_key)          # the flowgraph's variable IS the azimuth
        self._ps, self._direction_key = ps, direction_key
        self.plotter = plotter if plotter is not None else recording_plotter(parent)
        ps.subscribe(direction_key, self.update)             # a write from either side redraws
        self.set_direction(direction)
        self.plotter.set_text(text)
        self.plotter.set_text_visible(text_visible, True)
        if callback:
            ps.subscribe(direction_key, callback)            # last, so that construction does not fire it

    def update(self, *args):
        profile = pointer_profile(self[BEAM_AZM_KEY]) if self[BEAM_ENB_KEY] else []
        self.plotter.set_profile(key='1' + BEAM_AZM_KEY, color_spec=BEAM_COLOR_SPEC, fill=True, profile=profile)
        self.plotter.update()

    def set_direction(self, direction):
        if direction is None:
            self[BEAM_ENB_KEY] = False
            self.update()
        else:
            self[BEAM_ENB_KEY] = True
            self[BEAM_AZM_KEY] = direction                   # notifies ps's subscribers, update() among them

    def set_text(self, text):
        self.plotter.set_text(text)

    def set_text_visible(self, visible):
        self.plotter.set_text_visible(visible)


def strongest_direction(ang, lvl=None):
    """One item of `baz.music_doa`'s ports 0/1 -> the compass direction in degrees, or None when the item holds no
    estimate (all strengths 0: the block's unused slots are (angle 0, strength 0), lib/baz_music_doa.cc:95,146-155)."""
    ang = list(ang)
    if not ang:
        return None
    if lvl is None:
        return float(ang[0])
    lvl = list(lvl)
    best = max(range(len(ang)), key=lambda i: (lvl[i], -i))
    if not (lvl[best] > 0):
        return None
    return float(ang[best])
