"""Only the pieces of pcdet.models that SURVEY §8(f) lists as next on the hot path live here (the detectors themselves are
consumers of pcdet.ops and out of scope): backbones_3d.pfe.bev_grid_pooling — the bilinear BEV gather."""
