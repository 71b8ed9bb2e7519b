"""Import-path shim: ``lib.*`` resolves to the MI355X-native host helpers (same paths as the reference)."""
