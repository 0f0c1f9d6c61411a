"""falcon-r1cs witness engine for MI355X -- package directory (imported as ``falcon_r1cs_amd``)."""
