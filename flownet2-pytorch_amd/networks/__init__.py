"""Host-side mirror of the reference's ``networks`` package for the custom-layer hot path only:
``networks.correlation_package.correlation``, ``networks.resample2d_package.resample2d`` and
``networks.channelnorm_package.channelnorm`` (import paths used by the reference's
networks/FlowNetC.py:8 and models.py:9-10).  The conv sub-networks are out of scope."""
