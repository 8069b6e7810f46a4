"""`vidi.*` import paths of the reference CLI, resolved to the MI355X implementation."""
