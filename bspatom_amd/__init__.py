"""bspatom_amd -- MI355X-native drop-in for BspAtom's matrices.f90 + DSYGV hot path."""
__version__ = "0.1.0"
