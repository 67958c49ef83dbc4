from . import models
