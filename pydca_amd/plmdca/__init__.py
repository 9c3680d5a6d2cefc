"""Pseudolikelihood-maximisation DCA on MI355X (mirror of pydca/plmdca)."""
