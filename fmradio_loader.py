"""Import helper: the package directory is `fm-radio_amd/` (hyphen), so load it under the module name
`fm_radio_amd` with importlib."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG = ROOT / "fm-radio_amd"


def load():
    if "fm_radio_amd" in sys.modules:
        return sys.modules["fm_radio_amd"]
    spec = importlib.util.spec_from_file_location("fm_radio_amd", PKG / "__init__.py", submodule_search_locations=[str(PKG)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["fm_radio_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
