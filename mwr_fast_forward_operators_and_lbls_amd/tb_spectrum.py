"""``TbCloudRTE`` -- the object-level call surface of the reference's hot path.

The reference drives pyrtlib like this (python_src/proc/PyRTlib_processing.py:123-127)::

    rte = TbCloudRTE(z_in[::-1], p_in[::-1], t_in[::-1], rh_in[::-1], frqs, ang)
    rte.init_absmdl(mdl)
    rte.satellite = False # downwelling!!!
    df_from_ground = rte.execute()
    tbs[i,:,k,j] = df_from_ground["tbtotal"].values

Changing ``from pyrtlib.tb_spectrum import TbCloudRTE`` (:28) to
``from mwr_fast_forward_operators_and_lbls_amd.tb_spectrum import TbCloudRTE`` keeps that code
running, with ``execute()`` evaluated by the HIP library (include/mwrt.h ``mwrt_tb_batch``).
Same constructor argument meaning (z km ascending, p hPa, T K, rh fraction, frq GHz, ELEVATION
angles in degrees), same ``init_absmdl`` names, same DataFrame columns and row order
(angle-major, frequency-minor) as pyrtlib [EXT].

Scope: ground-based (downwelling); clear sky and plane-parallel by default like the
reference's calls.  Opt-in, as in pyrtlib: ``cloudy=True`` + ``init_cloudy(cldh, denice, denliq)`` (cloud
liquid / ice absorption) and ``ray_tracing=True`` (spherical refracted slant paths).  What is still
outside (upwelling, ozone, amu perturbation) raises ``NotImplementedError`` -- never a silent approximation.
pyrtlib keeps the model in process-global class state (why the reference re-issues
``init_absmdl`` before every ``execute``, :124,:132,:140,:148); here it is per instance.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import spectroscopy
from . import _native

DATAFRAME_COLUMNS = ["tbtotal", "tbatm", "tmr", "tmrcld", "tauwet", "taudry", "tauliq", "tauice"]


class TbCloudRTE(object):
    """Drop-in for ``pyrtlib.tb_spectrum.TbCloudRTE`` on the reference's path."""

    def __init__(self, z, p, t, rh, frq, angles: Optional[np.ndarray] = np.array([90.]),
                 o3n=None, amu=None, absmdl: Optional[str] = '', ray_tracing: Optional[bool] = False,
                 from_sat: Optional[bool] = True, cloudy: Optional[bool] = False):
        self.z = np.asarray(z, dtype=np.float64)
        self.p = np.asarray(p, dtype=np.float64)
        self.tk = np.asarray(t, dtype=np.float64)
        self.rh = np.asarray(rh, dtype=np.float64)
        self.frq = np.atleast_1d(np.asarray(frq, dtype=np.float64))
        self.angles = np.atleast_1d(np.asarray(angles, dtype=np.float64))
        if not (self.z.shape == self.p.shape == self.tk.shape == self.rh.shape) or self.z.ndim != 1:
            raise ValueError("z, p, t, rh must be 1-D arrays of equal length")
        self.nl = len(self.z)
        self.nf = len(self.frq)
        self.nang = len(self.angles)
        self.o3n = o3n
        self.amu = amu
        self.ray_tracing = ray_tracing
        self._satellite = from_sat
        self.cloudy = cloudy
        self._absmdl = absmdl
        self._tables = None
        self.cldh = None
        self.denliq = None
        self.denice = None
        if absmdl:
            self.init_absmdl(absmdl)

    # -- pyrtlib-compatible attributes ---------------------------------------------------------
    @property
    def satellite(self) -> bool:
        return self._satellite

    @satellite.setter
    def satellite(self, sat: bool) -> None:
        if not isinstance(sat, bool):
            raise ValueError("Please enter a valid value for satellite")
        self._satellite = sat

    def init_absmdl(self, absmdl: str):
        """Select the absorption model ("R20", "R24", "R17", "R98": PyRTlib_processing.py:122-146)."""
        self._tables = spectroscopy.get_model(absmdl)     # ValueError for an unknown name
        self._absmdl = absmdl

    def set_amu(self, amu) -> None:
        raise NotImplementedError("spectroscopic-uncertainty perturbation (amu) is outside the hot path")

    def init_cloudy(self, cldh, denice, denliq) -> None:
        """pyrtlib's signature: cloud base / top heights ``cldh`` (2, ncld) [km], ice and liquid density
        profiles [g m-3] on the instance's levels.  Only read when ``cloudy`` is True (opt-in: the reference
        runs clear sky, old_processing.py:558-563).  ``cldh`` only feeds pyrtlib's cloud radiating temperature
        (``tmrcld``), which is not evaluated here (the column stays 0)."""
        denice = np.asarray(denice, dtype=np.float64)
        denliq = np.asarray(denliq, dtype=np.float64)
        if denice.shape != self.z.shape or denliq.shape != self.z.shape:
            raise ValueError("denice and denliq must have one value per level")
        self.cldh = np.asarray(cldh, dtype=np.float64)
        self.denice = denice
        self.denliq = denliq

    # -- the hot path -----------------------------------------------------------------------------
    def execute(self, only_bt: Optional[bool] = True):
        """Run the RTE; returns a DataFrame with ``tbtotal`` etc. (and the layer dict if not only_bt)."""
        import pandas as pd

        if self._tables is None:
            raise ValueError("absorption model not set: call init_absmdl(<model>) first")
        if self._satellite:
            raise NotImplementedError("upwelling (satellite=True) is outside the hot path; the reference "
                                      "sets rte.satellite = False (PyRTlib_processing.py:125)")
        o3n = None
        if self.o3n is not None:
            # pyrtlib: o3n = ozone number density [molecules m-3] per level; O3AbsModel joins the dry absorption.  The
            # mechanism is here (include/mwrt.h mwrt_model_desc.n_x), the line list is not: it could not be restated
            # offline, so a model without one still refuses (tools/export_pyrtlib_tables.py --o3 +
            # spectroscopy.register_model / ModelTables.with_extra_lines install one).
            if self._tables.n_x == 0:
                raise NotImplementedError(f"ozone profile (o3n): model {self._absmdl!r} carries no O3 line table -- the list "
                                          "could not be restated offline; register one (ModelTables.with_extra_lines)")
            o3n = np.asarray(self.o3n, dtype=np.float64)
            if o3n.shape != self.z.shape:
                raise ValueError("o3n must have one value per level")
            o3n = o3n[None, :]
        denliq = denice = None
        if self.cloudy:
            if self.denliq is None:
                raise AttributeError("Set cloudy to True before running init_cloudy()")     # pyrtlib's wording
            denliq, denice = self.denliq[None, :], self.denice[None, :]

        z, p, t, rh = (np.ascontiguousarray(a, dtype=np.float64)[None, :] for a in (self.z, self.p, self.tk, self.rh))
        tb, valid, ex = _native.default_context().tb_batch(self._tables, z, p, t, rh, self.frq, self.angles, extras=True,
                                                           denliq=denliq, denice=denice,
                                                           ray_tracing=bool(self.ray_tracing), o3n=o3n)
        if valid[0] == 2:
            # pyrtlib raises inside RTEquation.exponential_integration on negative absorption
            raise ValueError("Error encountered in exponential_integration")
        if valid[0] == 3:
            raise ValueError("RayTrac_xxx: Ducting")
        n = self.nang * self.nf
        zeros = np.zeros(n)
        df = pd.DataFrame({'tbtotal': tb[0].reshape(n), 'tbatm': ex["tbatm"][0].reshape(n),
                           'tmr': ex["tmr"][0].reshape(n), 'tmrcld': zeros,
                           'tauwet': ex["tauwet"][0].reshape(n), 'taudry': ex["taudry"][0].reshape(n),
                           'tauliq': ex["tauliq"][0].reshape(n), 'tauice': ex["tauice"][0].reshape(n)})
        if only_bt:
            return df
        if self.ray_tracing:
            raise NotImplementedError("per-angle layer optical depths are not returned with ray_tracing=True "
                                      "(use only_bt=True)")
        # per-angle layer optical depths, shaped like pyrtlib's (nf, nang, nl) arrays
        amass = 1.0 / np.sin(self.angles * np.pi / 180)
        taulay = ex["taulay"][0][:, None, :] * amass[None, :, None]
        return df, {'taulay': taulay}
