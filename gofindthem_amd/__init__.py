"""gofindthem_amd -- MI355X-native implementation of gofindthem's ProcessText hot path.

The product is the C-ABI library built from gofindthem_amd/csrc (include/gft.h); this package is the thin
host-side mirror of the reference's finder API on top of it.  See DESIGN.md.
"""
