"""The callers of the generation path on the evaluation side: predictors (lcm/evaluation/predictors in the reference)."""
