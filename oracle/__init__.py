"""CPU oracles for the PnP hot path: test infrastructure, never imported by the product."""
