"""Mirror of the slice of the reference's ``common`` package that sits on the hot path (runner.py)."""
