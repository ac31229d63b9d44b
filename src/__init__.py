"""Drop-in `src` package: same dotted module paths as the reference (malteprinzler/diner) so that
configs and checkpoints that name `src.models.*` classes resolve to the MI355X-native implementation."""
