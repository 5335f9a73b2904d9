"""Drop-in stand-in for the `mitsuba` Python module on the radar hot path.

    from beifong_amd import mitsuba
    mitsuba.set_variant('scalar_rgb')
    from beifong_amd.mitsuba.core.xml import load_file
    scene = load_file('trans_rad.xml', spp=16)
    scene.integrator().render(scene, scene.sensors()[0])
    bmp = np.array(scene.sensors()[0].film().bitmap(raw=True))     # [H, W, 5 + aovs]

mirrors python_scripts/trans_rad.py:8-41 of the reference with a one-line import
change.  Everything below the Python surface is C++ (beifong_amd/host) and HIP
(beifong_amd/csrc); see INTEGRATION.md.
"""
from . import _host


def set_variant(name):
    _host.check(_host.lib().bfh_set_variant(name.encode()))


def variant():
    return _host.lib().bfh_variant().decode()


def variants():
    return ["scalar_rgb", "scalar_mono", "scalar_spectral"]


def set_gpu_count(n):
    """Ours, not the reference's: shard every render() / receive() over the first n GPUs of this process (sample shards +
    one RCCL all-reduce of the histogram, bf_render_sharded; `bfrender --gpus N` is the same switch)."""
    _host.check(_host.lib().bfh_set_gpu_count(int(n)))


def gpu_count():
    return _host.lib().bfh_gpu_count()
