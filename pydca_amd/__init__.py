"""pydca_amd -- MI355X-native compute core for pydca's plmDCA and mfDCA `compute_fn` paths.

Host-side mirror of the reference's interface for the hot path only:
  pydca_amd.plmdca.plmdca.PlmDCA                  <-> pydca/plmdca/plmdca.py
  pydca_amd.meanfield_dca.meanfield_dca.MeanFieldDCA <-> pydca/meanfield_dca/meanfield_dca.py
  pydca_amd.meanfield_dca.msa_numerics             <-> pydca/meanfield_dca/msa_numerics.py
  pydca_amd.plmdca_main / mfdca_main               <-> the `plmdca` / `mfdca` command lines
All numerics run in libdca_hip.so (include/dca_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"
