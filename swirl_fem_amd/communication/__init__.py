"""Setup-time collectives (mirror of `swirl_fem/communication/`)."""
