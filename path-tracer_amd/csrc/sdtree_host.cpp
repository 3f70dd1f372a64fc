// sdtree_host.cpp — host-side SD-tree refinement (filled in with the guided integrator)
