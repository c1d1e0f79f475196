"""Boundary -> regular-grid evaluators with the class API of ipde/grid_evaluators/."""
