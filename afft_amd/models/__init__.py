"""Mirrors of the reference's ``models`` package (same module names, class names and parameter names)."""
