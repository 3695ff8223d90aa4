"""Drop-in `diffuser` package surface for the hot path (MI355X-native).  Only the modules on the path named by
BASELINE.json's north_star exist here; simulator / trainer glue stays with the user's tree."""
