"""ruart_amd: MI355X-native hot path of RUArt (BERT encoder -> SDNet trunk -> answer scores)."""
