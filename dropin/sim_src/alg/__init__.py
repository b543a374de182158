"""Overlay of `sim_src.alg`: only `mmw` is replaced; the rest resolves from the reference tree."""
import pkgutil

__path__ = pkgutil.extend_path(__path__, __name__)


class alg_interface:
    """Same two no-op hooks as the reference's `sim_src/alg/__init__.py:1-5` (binary_search_relaxation
    imports this name from the package)."""

    def set_state(self, state):
        pass

    def slv(self):
        pass
