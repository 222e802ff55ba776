"""
Brain-specific variant -- counterpart of glimslib/simulation/simulation_tumor_growth_brain.py:8-152: the same
equations as TumorGrowth with per-tissue constants for the subdomains named 'CSF', 'WM', 'GM', 'Ventricles'
(and optionally 'outside'), no diffusion / proliferation in CSF and ventricles (:93-104).

Reference quirk q1: with an 'outside' subdomain the reference calls the non-existent
``mrd.compute_expansion`` (:75) and raises AttributeError.  The intended term -- the growth-induced strain with the
global coupling, as in simulation_tumor_growth_brain_quad.py:76 -- is what is implemented here, with the hard-wired
'outside' material E = 10e3, nu = 0.45 (:37-38).

The reference's brain form names its source term ``rd_source_term`` (:46, :104) and carries neither a body force nor
von Neumann terms (':105 No Von Neumann BC implemented here').  Here ``rd_source_term`` is read first (``source_term`` is
accepted as a synonym); body force and Neumann data are honoured when given -- an extension over the reference's brain
form, identical to it when they are absent, as in every bundled script.
"""
from __future__ import annotations

import numpy as np

from ..fenics_local import Constant
from .simulation_tumor_growth import TumorGrowth
from . import config


class TumorGrowthBrain(TumorGrowth):

    def _define_model_params(self):
        self.required_params = ['E_GM', 'E_WM', 'E_CSF', 'E_VENT',
                                'nu_GM', 'nu_WM', 'nu_CSF', 'nu_VENT',
                                'D_GM', 'D_WM',
                                'rho_GM', 'rho_WM',
                                'coupling']
        self.optional_params = []

    def _rd_source(self):
        src = getattr(self, 'rd_source_term', None)
        return src if src is not None else getattr(self, 'source_term', None)

    def _material_tables(self, n_labels):
        p = self.params
        f = lambda v: float(v) if isinstance(v, Constant) or not hasattr(v, '__len__') else float(np.asarray(v))
        tissues = {
            'CSF': dict(D=0.0, rho=0.0, E=f(p.E_CSF), nu=f(p.nu_CSF)),
            'WM': dict(D=f(p.D_WM), rho=f(p.rho_WM), E=f(p.E_WM), nu=f(p.nu_WM)),
            'GM': dict(D=f(p.D_GM), rho=f(p.rho_GM), E=f(p.E_GM), nu=f(p.nu_GM)),
            'Ventricles': dict(D=0.0, rho=0.0, E=f(p.E_VENT), nu=f(p.nu_VENT)),
            'outside': dict(D=0.0, rho=0.0, E=10E3, nu=0.45),
        }
        t = dict(D=np.zeros(n_labels), rho=np.zeros(n_labels), gamma=np.zeros(n_labels),
                 E=np.ones(n_labels), nu=np.full(n_labels, 0.3))
        present = np.unique(self._labels())
        known = {}
        for name, vals in tissues.items():
            tid = self.subdomains.tissue_name_id_map.get(name) if hasattr(self.subdomains, 'tissue_name_id_map') \
                else {v: k for k, v in getattr(self.subdomains, 'tissue_id_name_map', {}).items()}.get(name)
            if tid is None:
                if name != 'outside':
                    self.logger.error("Subdomain '%s' does not exist" % name)
                continue
            known[tid] = name
            if tid < n_labels:
                for k in ('D', 'rho', 'E', 'nu'):
                    t[k][tid] = vals[k]
                t['gamma'][tid] = f(p.coupling)
        missing = [int(l) for l in present if int(l) not in known]
        if missing:
            raise ValueError("cells carry tissue ids %s that TumorGrowthBrain has no material for "
                             "(expected names CSF/WM/GM/Ventricles[/outside] in domain_names)" % missing)
        return t

    def run_for_adjoint(self, parameters, output_dir=config.output_dir_simulation_tmp):
        """:127-145 -- (D_WM, D_GM, rho_WM, rho_GM, coupling)"""
        self.params.D_WM, self.params.D_GM = parameters[0], parameters[1]
        self.params.rho_WM, self.params.rho_GM = parameters[2], parameters[3]
        self.params.coupling = parameters[4]
        self.run(keep_nth=1, save_method=None, clear_all=False, plot=False, output_dir=output_dir)
        return self.solution

    def init_postprocess(self, output_dir=config.output_dir_simulation_tmp):
        """:148-152"""
        from ..simulation_helpers.postprocess import PostProcessTumorGrowthBrain
        if self._backend is None:
            raise RuntimeError("init_postprocess needs a finished run()")
        labels = self._labels()
        t = self._material_tables(int(labels.max()) + 1)
        self.postprocess = PostProcessTumorGrowthBrain(self.results, self.params, output_dir=output_dir,
                                                       backend=self._backend, tables=t, labels=labels)
        self.postprocess.map_params()
        return self.postprocess
