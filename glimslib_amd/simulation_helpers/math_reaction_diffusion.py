"""Same closed form as glimslib/simulation_helpers/math_reaction_diffusion.py:2-3 (works on floats and arrays)."""


def compute_growth_logistic(conc, prolif_rate, conc_max):
    return prolif_rate * conc * (1 - conc / conc_max)
