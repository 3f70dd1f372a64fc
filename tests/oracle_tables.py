"""The CIE tables the oracle compiles in (oracle/cie_data.inc), parsed for numpy-side checks."""
import re
from pathlib import Path

import numpy as np

_INC = Path(__file__).resolve().parent.parent / "oracle" / "cie_data.inc"


def _body(name):
    text = _INC.read_text().replace("\\\n", " ")
    m = re.search(r"#define " + name + r"\s+(.*)", text)
    return m.group(1)


def cie_d65():
    return np.array([float(t.rstrip("f")) for t in _body("PT_CIE_D65_VALUES").split(",") if t.strip()], dtype=np.float32)


def cie_xyz():
    rows = re.findall(r"\{([^{}]+)\}", _body("PT_CIE_XYZ_VALUES"))
    return np.array([[float(v) for v in r.split(",")] for r in rows], dtype=np.float64)
