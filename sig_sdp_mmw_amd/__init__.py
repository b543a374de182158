"""MI355X-native MMW SDP solver behind the `mmw` solver-class surface of zhouyou-gu/sig-sdp-mmw."""
__version__ = "0.1.0"
