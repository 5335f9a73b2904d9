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


def dict_to_xml(d, name=None, indent=0):
    """Serialise a Mitsuba scene dictionary (load_dict format) to scene XML."""
    pad = "  " * indent
    d = getattr(d, "_dict", d)          # objects returned by load_dict re-serialise from their source dict
    if not isinstance(d, dict) or "type" not in d:
        raise ValueError("load_dict: every object needs a 'type'")
    typ = d["type"]
    nm = f' name="{_esc(name)}"' if name and not str(name).startswith("_arg_") else ""
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
            out += dict_to_xml(vv, k, indent + 1)
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
        # single objects cannot be instantiated standalone by the scene loader's
        # entry point; keep the dict so that a parent load_dict can embed it
        class Deferred:
            pass

        o = Deferred()
        o._dict = dict(d)
        return o
    o = load_string(text, base_dir=base_dir)
    o._dict = dict(d)
    return o
