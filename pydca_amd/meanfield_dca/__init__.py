"""Mean-field DCA on MI355X (mirror of pydca/meanfield_dca)."""
