"""Drop-in overlay of the reference's `sim_src` package.

Put this directory FIRST on PYTHONPATH, the reference checkout after it:

    PYTHONPATH=/path/to/this/repo/dropin:/path/to/this/repo:/path/to/sig-sdp-mmw  python sim_script/pd_mmw_template.py

`from sim_src.alg.mmw import mmw` (sim_script/pd_mmw_template.py:12) then resolves to the MI355X solver,
every other `sim_src.*` module (env, util, binary_search_relaxation, ...) still comes from the reference
tree, unchanged.  Nothing of the reference is copied here.
"""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)
