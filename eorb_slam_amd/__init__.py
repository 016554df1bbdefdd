"""MI355X-native event-frame front end (accumulate -> ORB extract -> Hamming match)."""
__version__ = "0.1.0"
