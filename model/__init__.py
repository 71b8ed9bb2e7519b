"""Import-path shim: ``model.*`` resolves to the MI355X-native modules (same paths as the reference)."""
