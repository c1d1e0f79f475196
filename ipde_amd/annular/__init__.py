"""Annular (Chebyshev x Fourier) solvers — host-side mirror of ipde/annular/."""
