import enum


class DrawReason(enum.Enum):
    """Why a frame is being drawn (same members/values as reference src/topsy/drawreason.py)."""
    INITIAL_UPDATE = 1
    CHANGE = 2
    REFINE = 3
    PRESENTATION_CHANGE = 4
    EXPORT = 5
