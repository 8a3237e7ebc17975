"""MI355X-native training hot path of EndoscopyDepthEstimation-Pytorch."""
