"""Local-only PathManager."""
import os


class PathHandler:
    pass


class HTTPURLHandler(PathHandler):
    pass


class OneDrivePathHandler(PathHandler):
    pass


class PathManager:
    def open(self, path, mode="r", **kw):
        return open(path, mode)

    def isfile(self, path):
        return os.path.isfile(path)

    def exists(self, path):
        return os.path.exists(path)

    def get_local_path(self, path, **kw):
        return path

    def register_handler(self, handler, **kw):
        pass
