"""mitsuba.core.xml: load_file / load_string / load_dict
(src/libcore/xml.cpp, src/python/python/xml.py)."""
import os

import ctypes as C
import numpy as np

from .. import _host
from ...scenedesc import Transform4f as _T

_OBJECT_TAGS = {
    "scene": "scene",
}
# plugin type -> XML tag (class aliases, xml.cpp:153-161)
_PLUGIN_TAG = {
    "path": "integrator", "pathlength": "integrator", "pathtime": "integrator", "range": "integrator", "time": "integrator",
    "pathtimefrequency": "integrator", "phase": "integrator", "rectangle": "shape", "obj": "shape", "ply": "shape", "diffuse": "bsdf",
    "twosided": "bsdf", "roughconductor": "bsdf", "spot": "emitter", "point": "emitter", "area": "emitter",
    "areatransmitter": "transmitter", "wignertransmitter": "transmitter", "phasedtransmitter": "transmitter",
    "fluxmeter": "sensor", "irradiancemeter": "sensor", "radiancemeter": "sensor", "perspective": "sensor",
    "omnidirectional": "receiver", "wignerreceiver": "receiver", "phasedreceiver": "receiver", "hdrfilm": "film", "hdradc": "adc",
    "box": "rfilter", "tent": "rfilter", "gaussian": "rfilter", "mitchell": "rfilter", "catmullrom": "rfilter", "lanczos": "rfilter",
    "independent": "sampler",
}


def load_file(path, **kwargs):
    n, keys, vals = _host.params(kwargs)
    out = C.c_void_p()
    _host.check(_host.lib().bfh_load_file(str(path).encode(), n, keys, vals, C.byref(out)))
    return _host.wrap(out)


def load_string(text, base_dir=".", **kwargs):
    n, keys, vals = _host.params(kwargs)
    out = C.c_void_p()
    _host.check(_host.lib().bfh_load_string(text.encode(), str(base_dir).encode(), n, keys, vals, C.byref(out)))
    return _host.wrap(out)


def _esc(s):
    return str(s).replace("&", "&amp;").replace('"', "&quot;").replace("<", "&lt;")


def _fmt(x):
    return repr(float(np.float32(x))) if not isinstance(x, (int, np.integer)) else str(int(x))


def dict_to_xml(d, name=None, indent=0, _seen=None):
    """Serialise a Mitsuba scene dictionary (load_dict format) to scene XML.  An object with an "id" that occurs several times
    (the same load_dict result embedded in several parents: one shared instance in the reference) is written once and
    referenced afterwards."""
    pad = "  " * indent
    if _seen is None:
        _seen = set()
    d = getattr(d, "_dict", d)          # objects returned by load_dict re-serialise from their source dict
    if not isinstance(d, dict) or "type" not in d:
        raise ValueError("load_dict: every object needs a 'type'")
    typ = d["type"]
    nm = f' name="{_esc(name)}"' if name and not str(name).startswith("_arg_") else ""
    if typ != "ref" and "id" in d:
        if d["id"] in _seen:
            return f'{pad}<ref id="{_esc(d["id"])}"{nm}/>\n'
        _seen.add(d["id"])
    if typ == "ref":
        return f'{pad}<ref id="{_esc(d["id"])}"{nm}/>\n'
    if typ == "spectrum":
        v = d["value"]
        if isinstance(v, (list, tuple)):
            v = ", ".join(f"{_fmt(w)}:{_fmt(x)}" for w, x in v)
        else:
            v = _fmt(v)
        return f'{pad}<spectrum{nm} value="{v}"/>\n'
    if typ == "rgb":
        v = d["value"]
        v = ", ".join(_fmt(x) for x in v) if isinstance(v, (list, tuple, np.ndarray)) else _fmt(v)
        return f'{pad}<rgb{nm} value="{v}"/>\n'
    tag = "scene" if typ == "scene" else _PLUGIN_TAG.get(typ)
    if tag is None:
        raise ValueError(f"load_dict: plugin type '{typ}' is not part of the radar path")
    ida = f' id="{_esc(d["id"])}"' if "id" in d else ""
    out = f'{pad}<scene version="2.1.0">\n' if tag == "scene" else f'{pad}<{tag} type="{typ}"{ida}{nm}>\n'
    for k, v in d.items():
        if k in ("type", "id"):
            continue
        vv = getattr(v, "_dict", v)
        if isinstance(vv, dict):
            out += dict_to_xml(vv, k, indent + 1, _seen)
        elif isinstance(v, _T):
            m = " ".join(_fmt(x) for x in np.asarray(v.matrix, dtype=np.float32).reshape(16))
            out += f'{pad}  <transform name="{_esc(k)}"><matrix value="{m}"/></transform>\n'
        elif isinstance(v, (bool, np.bool_)):
            out += f'{pad}  <boolean name="{_esc(k)}" value="{"true" if v else "false"}"/>\n'
        elif isinstance(v, (int, np.integer)):
            out += f'{pad}  <integer name="{_esc(k)}" value="{int(v)}"/>\n'
        elif isinstance(v, (float, np.floating)):
            out += f'{pad}  <float name="{_esc(k)}" value="{_fmt(v)}"/>\n'
        elif isinstance(v, str):
            out += f'{pad}  <string name="{_esc(k)}" value="{_esc(v)}"/>\n'
        elif isinstance(v, (list, tuple, np.ndarray)) and len(v) == 3:
            out += f'{pad}  <vector name="{_esc(k)}" value="{", ".join(_fmt(x) for x in v)}"/>\n'
        else:
            raise ValueError(f"load_dict: unsupported value for '{k}': {type(v)}")
    out += f"{pad}</{tag}>\n"
    return out


def load_dict(d, base_dir="."):
    """src/python/python/xml.py load_dict: note that a Transform4f given as a
    matrix loses its analytic inverse (Transform(matrix) inverts numerically,
    transform.h), exactly as in the reference."""
    text = dict_to_xml(d)
    if _PLUGIN_TAG.get(d.get("type")) == "rfilter":
        # a reconstruction filter is complete on its own (eval / eval_discretized / radius: src/rfilters/tests/test_rfilter.py)
        o = load_string(text.replace("<rfilter ", '<rfilter version="2.1.0" ', 1), base_dir=base_dir)
        o._dict = dict(d)
        return o
    if d.get("type") != "scene":
        # an object outside a scene is kept as its dictionary until a scene that embeds it is loaded; from then on it stands
        # for its instance in that scene (the reference's scripts keep using such objects: animated_trans_rad.py:316-377 renders
        # `scene.integrator().render(scene, sen)` and reads `film.bitmap(raw=True)` with `sen`, `film` made by load_dict)
        o = Deferred()
        o._dict = dict(d)
        return o
    o = load_string(text, base_dir=base_dir)
    o._dict = dict(d)
    _bind(d, o)
    return o


class Deferred:
    """Result of load_dict for an object that is not a scene (see load_dict)."""
    _target = None

    def _resolve(self):
        if self._target is None:
            raise AttributeError("this object was made by load_dict outside a scene: it becomes usable once a scene that "
                                 "contains it has been loaded with load_dict")
        return self._target() if callable(self._target) else self._target

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._resolve(), name)


def _bind(scene_dict, scene):
    """Point the Deferred objects embedded in a scene dictionary at their instances in the loaded scene: sensors and
    receivers in the order of the dictionary, their film / adc / sampler children through them."""
    sensors, receivers = scene.sensors(), scene.receivers()
    si = ri = 0
    for v in scene_dict.values():
        dd = getattr(v, "_dict", v)
        if not isinstance(dd, dict):
            continue
        tag = _PLUGIN_TAG.get(dd.get("type"))
        # a shape may carry the sensor / receiver (rectangle with a child endpoint)
        holders = [(v, dd)] + [(c, getattr(c, "_dict", c)) for c in dd.values() if isinstance(getattr(c, "_dict", c), dict)] \
            if tag == "shape" else [(v, dd)]
        for obj, od in holders:
            t = _PLUGIN_TAG.get(od.get("type"))
            inst = None
            if t == "sensor" and si < len(sensors):
                inst, si = sensors[si], si + 1
            elif t == "receiver" and ri < len(receivers):
                inst, ri = receivers[ri], ri + 1
            if inst is None:
                continue
            if isinstance(obj, Deferred):
                obj._target = inst
            for c in od.values():
                if isinstance(c, Deferred):
                    ct = _PLUGIN_TAG.get(c._dict.get("type"))
                    if ct == "film":
                        c._target = inst.film
                    elif ct == "adc":
                        c._target = inst.adc
                    elif ct == "sampler":
                        c._target = inst.sampler
