"""Minimal reference-compatible ``deepclustering2`` namespace: only what the semi-supervised train-step
path imports.  The reference vendors the full package as a wheel; its arithmetic
pieces used here (KL_div, one-hot helpers, flips, Dice, Adam) are re-implemented against the MI355X
kernels or as thin host logic.  Import paths match the wheel so ``semi_seg`` reads like the reference.
"""
