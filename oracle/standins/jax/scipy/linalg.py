from scipy.linalg import cho_solve, cho_factor, solve_triangular  # noqa: F401
