"""Opt-in shim: put this directory on PYTHONPATH (``PYTHONPATH=<repo>/shim:<repo>``) and ``import clipcap`` / ``python -m clipcap.train``
resolve to clipcap_amd (clipcap_amd.install_as_clipcap replaces this module in sys.modules).  Not on the path by default, so an
installed reference ``clipcap`` is never shadowed by accident."""
import clipcap_amd

clipcap_amd.install_as_clipcap(force=True)
