class NeighborList:
    pass


class NewPrimitiveNeighborList:
    pass


class PrimitiveNeighborList:
    pass
