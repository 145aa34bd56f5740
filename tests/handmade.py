"""Hand-built scene descriptions shared by the CPU and the GPU tests."""


def nested_fold_scene(capi):
    """Three spheres behind one another in three leaf boxes, two of them inside an inner
    box: top level [leaf A, inner [leaf B, leaf C]].  With these materials the per-level
    colour fold of raytracing.h:85-93 differs from one running fold over all leaves.
    Returns (description, objects that must stay alive while it is used)."""
    mats = (capi.Material * 3)()
    for k, dif in enumerate([(0.20, 0.20, 0.20), (0.02, 0.60, 0.60), (0.25, 0.25, 0.25)]):
        mats[k].ior, mats[k].roughness = 1.1, 0.9
        mats[k].specular[:] = (0.0, 0.0, 0.0)
        mats[k].diffuse[:] = dif
    prims = (capi.Prim * 3)()
    for k, z in enumerate([700.0, 500.0, 300.0]):
        prims[k].type = capi.RTX_PRIM_SPHERE
        prims[k].g[0:4] = (500.0, 500.0, z, 60.0)
        prims[k].material, prims[k].id = k, k
    nodes = (capi.Node * 4)()
    big_lb, big_rt = (0.0, 0.0, 0.0), (1000.0, 1000.0, 1000.0)
    for n in nodes:
        n.lb[:], n.rt[:] = big_lb, big_rt
    nodes[0].skip, nodes[0].leaf = 1, 0
    nodes[1].skip, nodes[1].leaf = 4, capi.RTX_NONE
    nodes[2].skip, nodes[2].leaf = 3, 1
    nodes[3].skip, nodes[3].leaf = 4, 2
    leaves = (capi.Leaf * 3)()
    for k in range(3):
        leaves[k].first, leaves[k].count = k, 1
    d = capi.SceneDesc()
    d.n_nodes, d.n_leaves, d.n_prims, d.n_materials = 4, 3, 3, 3
    d.nodes, d.leaves, d.prims, d.materials = nodes, leaves, prims, mats
    d.light.position[:] = (250.0, 250.0, 800.0)
    d.light.color[:] = (255.0, 255.0, 255.0)
    return d, (mats, prims, nodes, leaves)
