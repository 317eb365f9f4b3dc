"""Import-path alias of the reference evaluation package."""
