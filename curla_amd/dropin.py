"""Zero-edit binding for the reference's drivers.

``train.py`` / ``eval.py`` / ``plot_tsne/*.py`` reach the learner path through four top-level module names
(train.py:20-25, eval.py:15-19): ``utils`` (FrameStack, ReplayBuffer, eval_mode, set_seed_everywhere, make_dir),
``augmentations`` (make_augmentor), ``curl_sac`` (CurlSacAgent) and, from curl_sac itself, ``encoder``.
curla_amd's modules carry the same names, so the drop-in is one of

    import curla_amd.utils as utils                    # edit the three import lines ...
    from curla_amd.augmentations import make_augmentor
    from curla_amd.curl_sac import CurlSacAgent

or, without touching the reference's files, registering the aliases before they are imported:

    python -c "import curla_amd.dropin as d, runpy; d.install(); runpy.run_path('train.py', run_name='__main__')"

``install()`` puts curla_amd's modules into ``sys.modules`` under the reference's names; ``uninstall()`` removes
exactly what it put there.  Nothing else is patched.
"""
import importlib
import sys

ALIASES = {
    "utils": "curla_amd.utils",                   # utils.py
    "augmentations": "curla_amd.augmentations",   # augmentations.py
    "curl_sac": "curla_amd.curl_sac",             # curl_sac.py
    "encoder": "curla_amd.encoder",               # encoder.py
}

_installed = {}


def install(force=False):
    """Register the aliases.  A different module already imported under one of the names (the reference's own
    ``utils`` on sys.path, say) is an error unless ``force`` -- silently shadowing half of the path would leave the
    driver with a mix of both implementations."""
    mods = {name: importlib.import_module(target) for name, target in ALIASES.items()}
    clash = [n for n, m in mods.items() if sys.modules.get(n) not in (None, m)]
    if clash and not force:
        raise ImportError("curla_amd.dropin.install(): %s already imported from elsewhere (%s); call install() before "
                          "the driver's imports, or pass force=True" %
                          (", ".join(clash), ", ".join(str(getattr(sys.modules[n], "__file__", "?")) for n in clash)))
    for name, mod in mods.items():
        sys.modules[name] = mod
        _installed[name] = mod
    return mods


def uninstall():
    for name, mod in list(_installed.items()):
        if sys.modules.get(name) is mod:
            del sys.modules[name]
        del _installed[name]
